#!/usr/bin/env python3
"""List VGPR / SGPR / spill / scratch / LDS figures of every kernel in the -save-temps .s files of the product build.
   python tools/kernel_resources.py [build_dir]     (exit code 1 if any kernel spills or uses scratch)"""
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

def main():
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "quantumattention_amd", "_build")
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
