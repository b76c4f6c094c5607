"""Turn the raw output of tools/profile_bench.sh (gpurun_out/prof_<tag>/) into the committed summaries:
  profiles/<name>/kernel_stats.csv   rocprofv3 --kernel-trace --stats (qattn kernels only)
  profiles/<name>/pmc_summary.json   mean counter value per dispatch and kernel, all PMC passes
  profiles/traffic.json              HBM bytes per attention launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)
usage: python tools/summarize_profile.py gpurun_out/prof_r02_c2 profiles/r02_c2 [B H S D [causal]]
(the traffic file is written as <dst>/traffic.json; for the headline shape also as profiles/traffic.json)
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _commit():
    import subprocess
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        return "unknown"


def _csrc_sha16():   # (bench.py csrc_sha16: the same bytes in the same order)
    # the collection run's own stamp (tools/profile_bench.sh), if it left one: the tree may have moved on since
    stamp = os.path.join(sys.argv[1], "csrc_sha16.txt")
    if os.path.exists(stamp) and open(stamp).read().strip():
        return open(stamp).read().strip()
    import hashlib
    h, d = hashlib.sha256(), os.path.join(ROOT, "quantumattention_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".inc")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


src, dst = sys.argv[1], sys.argv[2]
shape = tuple(int(x) for x in sys.argv[3:7]) if len(sys.argv) >= 7 else (4, 32, 4096, 128)
causal = len(sys.argv) >= 8 and sys.argv[7] not in ("0", "false")
os.makedirs(dst, exist_ok=True)
csv.field_size_limit(1 << 30)

newest = lambda paths: sorted(paths, key=os.path.getmtime)[-1:]   # gpurun merges runs into one directory: keep the latest of each pass
stats = newest(glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True))
if stats:
    with open(stats[0]) as f, open(os.path.join(dst, "kernel_stats.csv"), "w") as g:
        for i, line in enumerate(f):
            if i == 0 or "qattn::" in line:
                g.write(line)

summary = {}
for cc in [f for d in sorted(glob.glob(os.path.join(src, "pmc_*"))) if os.path.isdir(d)
           for f in newest(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))]:
    acc = {}
    with open(cc) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if "qattn::" not in name:
                continue
            key = name.split("(")[0][:90]
            acc.setdefault((key, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
    for (k, c), v in acc.items():
        summary.setdefault(k, {})[c] = sum(v) / len(v)
with open(os.path.join(dst, "pmc_summary.json"), "w") as f:
    json.dump(summary, f, indent=1)

attn = [k for k in summary if "attn_fwd_kernel" in k]
# the in-step instantiation (last template argument Q16 = true: it reads Q as bf16) when present, else the plain one
def _targs(k):   # template arguments of attn_fwd_kernel_v2<D, NW, QK, V, CAUSAL, TOKEN, BYTE, ABL, Q16, CHECK>
    return [a.strip() for a in k[k.index("<") + 1:k.rindex(">")].split(",")] if "<" in k and ">" in k else []
fused = [k for k in attn if "kernel_v2" in k and len(_targs(k)) >= 9 and _targs(k)[8] == "true"]
if attn:
    key = fused[0] if fused else attn[0]
    s = summary[key]
    # FETCH_SIZE / WRITE_SIZE count 64-byte... units of 1 KiB per the guide's rocprofv3 section; FETCH_SIZE under-reports by 2x on gfx950
    fetch = s["FETCH_SIZE"] * 1024 * 2
    write = s["WRITE_SIZE"] * 1024
    B, H, S, D = shape
    out = {
        "shape": {"B": B, "H": H, "S": S, "D": D, "causal": causal},
        "attn_fwd_hbm_bytes_per_launch": fetch + write,
        "fetch_bytes_x2_corrected": fetch,
        "write_bytes": write,
        "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section; %s/pmc_summary.json" % dst,
        "algorithmic_bytes": B * H * S * D * ((2 + 1 + 1 + 2) if fused else (3 + 2)),  # Q (bf16 or fp8) + K + V fp8 + O bf16
        "kernel": key,
        # which tree the counters belong to: bench.py reports whether the kernel sources have changed since (it cannot collect counters itself)
        "commit": _commit(),
        "csrc_sha16": _csrc_sha16(),
    }
    if causal:   # K/V re-streaming if no query block shared them = (#256-row blocks per head) x 1/2 x (K + V per head)
        out["kv_restream_bytes_without_l2_reuse"] = B * H * (S // 256) * (S * D * 2) // 2
    with open(os.path.join(dst, "traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    if shape == (4, 32, 4096, 128) and not causal and fused:   # the bench line's kernel only (not the token-wise / 16-bit profiles)
        with open(os.path.join(os.path.dirname(dst.rstrip("/")), "traffic.json"), "w") as f:
            json.dump(out, f, indent=1)
    print(json.dumps(out))
