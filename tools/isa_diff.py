#!/usr/bin/env python3
"""Compare the device assembly of two -save-temps builds instruction for instruction (comments, .file / .ident / .loc lines stripped:
label comments carry the mangled names of inlined functions, which change with any template signature).
   python tools/isa_diff.py <dir_before> [<dir_after> = quantumattention_amd/_build_temps]
Used in round 6 to show that taking the QATTN_DEV scaffolding out of the product sources left every kernel's ISA unchanged."""
import glob, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
before = sys.argv[1]
after = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "quantumattention_amd", "_build_temps")


def norm(path):
    out = []
    for line in open(path, errors="replace"):
        line = re.sub(r";.*$", "", line).rstrip()
        if line.strip() and not re.search(r"\.file|\.ident|\.loc\b", line):
            out.append(line)
    return out


bad = 0
for f in sorted(glob.glob(os.path.join(after, "*gfx950.s"))):
    g = os.path.join(before, os.path.basename(f))
    if not os.path.exists(g):
        print(f"{os.path.basename(f):60s} (no counterpart)")
        continue
    a, b = norm(f), norm(g)
    same = a == b
    bad += not same
    n_mfma = sum("v_mfma" in x for x in a)
    print(f"{os.path.basename(f):60s} {'IDENTICAL' if same else 'DIFFERS  '}  {len(a)} lines, {n_mfma} v_mfma")
sys.exit(1 if bad else 0)
