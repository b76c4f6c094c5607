#!/usr/bin/env python3
"""Compare two -save-temps builds KERNEL BY KERNEL, modulo scalar-register numbering:
   python tools/isa_funcs.py <dir_before> <dir_after> <unit> [<unit> ...]
For every function of `<unit>-hip-amdgcn-amd-amdhsa-gfx950.s` present in both builds: instruction lines (comments, .file / .loc / .ident and
the translation unit's __hip_cuid stripped; a trailing `, 0` template argument of attn_fwd_kernel_v2 -- the SV name tag of round 6 -- ignored),
with every `sN` / `s[a:b]` replaced by `s`: `same`, or the number of differing runs and the first few of them.
Used in round 6 to show that the DENSE units of the hand-scheduled kernel (csrc/qattn_attn.h QATTN_STRIDED16 = 0) still are the kernels of the
tree before strided views existed: what differs is the kernel-argument size (the parameter struct grew), one scalar constant for it -- hence
renumbered registers -- and a handful of instruction-selection differences in the block prologue (1-3 % of a fused kernel's lines; with
run-time strides in the same source: half of them)."""
import difflib
import os
import re
import sys


def funcs(path):
    out, cur = {}, None
    for line in open(path, errors="replace"):
        line = re.sub(r";.*$", "", line).rstrip()
        line = re.sub(r"__hip_cuid_\w+", "__hip_cuid", line).replace("ELi0EEEvNS_10AttnParamsE", "EEEvNS_10AttnParamsE")
        if not line.strip() or re.search(r"\.file|\.ident|\.loc\b", line):
            continue
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        if cur is not None and line.startswith("\t") and not line.startswith("\t."):
            line = re.sub(r"\bs\[\d+:\d+\]", "s[]", line)
            cur.append(re.sub(r"\.LBB\d+_", ".LBB_", re.sub(r"\bs\d+\b", "s", line)).strip())
    return out


before, after = sys.argv[1], sys.argv[2]
for unit in sys.argv[3:]:
    name = unit + "-hip-amdgcn-amd-amdhsa-gfx950.s"
    a, b = funcs(os.path.join(before, name)), funcs(os.path.join(after, name))
    print(f"== {unit}")
    for n in sorted(b):
        if n not in a:
            print(f"  new      {n[:120]}")
            continue
        if a[n] == b[n]:
            print(f"  same     {len(b[n]):6d} instructions  {n[:120]}")
            continue
        ops = [o for o in difflib.SequenceMatcher(None, a[n], b[n], autojunk=False).get_opcodes() if o[0] != "equal"]
        moved = sum(max(i2 - i1, j2 - j1) for _, i1, i2, j1, j2 in ops)
        print(f"  differs  {len(b[n]):6d} instructions, {len(ops)} runs / {moved} lines  {n[:120]}")
        for tag, i1, i2, j1, j2 in ops[:6]:
            print(f"      {tag:8s} {[x[:48] for x in a[n][i1:i2][:3]]} -> {[x[:48] for x in b[n][j1:j2][:3]]}")
