#!/usr/bin/env python3
"""List VGPR / SGPR / spill / scratch / LDS figures of every kernel in the -save-temps .s files of the product build.
   python tools/kernel_resources.py [build_dir]     (exit code 1 if any kernel spills or uses scratch)
   python tools/kernel_resources.py --loops <kernel name substring> [build_dir]
       per loop of that kernel: MFMA count, v_readlane / v_writelane (SGPR lane spills) and scratch accesses inside it"""
import os, re, subprocess, sys

def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return out[:len(names)]
    except Exception:
        return names

def parse(path):
    rows, cur = [], None
    for line in open(path, errors="replace"):
        m = re.match(r"\s+\.name:\s+(\S+)", line)
        if m and cur is not None and "name" not in cur:
            cur["name"] = m.group(1)
        if re.match(r"\s+- \.agpr_count:", line) or re.match(r"\s+- \.args:", line):
            cur = {}
            rows.append(cur)
        for key in ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"):
            m = re.match(r"\s+(?:- )?\.%s:\s+(\d+)" % key, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return [r for r in rows if "name" in r and "vgpr_count" in r]

def loops(d, pattern):
    for f in sorted(os.listdir(d)):
        if not f.endswith("gfx950.s"):
            continue
        text = open(os.path.join(d, f), errors="replace").read().split("\n")
        starts = [i for i, l in enumerate(text) if re.match(r"^_Z\w+:", l) and pattern in l]
        for st in starts:
            end = next(i for i in range(st, len(text)) if ".end_amdhsa_kernel" in text[i] or (i > st and re.match(r"^_Z\w+:", text[i])))
            body = text[st:end]
            labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
            seen = set()
            print("==", body[0][:140])
            for i, l in enumerate(body):
                m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
                if m and m.group(1) in labels and labels[m.group(1)] < i and (labels[m.group(1)], i) not in seen:
                    a = labels[m.group(1)]
                    seen.add((a, i))
                    seg = body[a:i + 1]
                    print("  loop lines %5d-%5d: mfma %3d  lane-spill ops %3d  scratch ops %3d" % (
                        a, i, sum("v_mfma" in x for x in seg), sum(("v_readlane" in x or "v_writelane" in x) for x in seg), sum("scratch_" in x for x in seg)))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--loops":
        d = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "quantumattention_amd", "_build_temps")
        loops(d, sys.argv[2])
        return 0
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "quantumattention_amd", "_build_temps")
    bad = 0
    for f in sorted(os.listdir(d)):
        if not f.endswith("gfx950.s"):
            continue
        rows = parse(os.path.join(d, f))
        names = demangle([r["name"] for r in rows])
        print(f"== {f}: {len(rows)} kernels")
        for r, n in zip(rows, names):
            sp = r.get("vgpr_spill_count", 0); sc = r.get("private_segment_fixed_size", 0)
            flag = "  <-- SPILL/SCRATCH" if (sp or sc) else ""
            bad += bool(sp or sc)
            n = re.sub(r"\(qattn::AttnParams.*", "", n)
            print(f"  vgpr {r['vgpr_count']:3d} agpr {r.get('agpr_count',0):3d} sgpr {r.get('sgpr_count',0):3d} spill {sp:3d} scratch {sc:4d} lds {r.get('group_segment_fixed_size',0):6d}  {n[:150]}{flag}")
    print(f"kernels with spills or scratch: {bad}")
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
