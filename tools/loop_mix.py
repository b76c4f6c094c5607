#!/usr/bin/env python3
"""Instruction mix of the MFMA loops of one kernel in the -save-temps assembly (build.py --save-temps).
   python tools/loop_mix.py <kernel name substring> [file.s]      e.g.  Lb0ELb0ELb1ELi0ELb1ELb1E  (non-causal, byte, fused Q, CHECK)"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1]
f = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "quantumattention_amd", "_build_temps", "qattn_attn_v2_e4m3-hip-amdgcn-amd-amdhsa-gfx950.s")
text = open(f, errors="replace").read().split("\n")
starts = [i for i, l in enumerate(text) if re.match(r"^_Z\w+:", l) and pat in l]
for st in starts:
    end = next(i for i in range(st + 1, len(text)) if ".end_amdhsa_kernel" in text[i] or re.match(r"^_Z\w+:", text[i]))
    body = text[st:end]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    print("==", body[0][:120])
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    for a, b in loops:
        seg = [x.strip() for x in body[a:b + 1] if x.startswith("\t") and not x.strip().startswith((";", "."))]
        n_mfma = sum("v_mfma" in x for x in seg)
        if n_mfma < 16 or n_mfma > 24:
            continue
        cls = {"mfma": 0, "valu": 0, "salu": 0, "ds_read": 0, "ds_write": 0, "vmem": 0, "waitcnt": 0, "nop": 0, "lane": 0, "other": 0}
        for x in seg:
            op = x.split()[0]
            if "v_mfma" in op: cls["mfma"] += 1
            elif op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): cls["lane"] += 1
            elif op.startswith("v_"): cls["valu"] += 1
            elif op == "s_waitcnt": cls["waitcnt"] += 1
            elif op == "s_nop": cls["nop"] += 1
            elif op.startswith("s_"): cls["salu"] += 1
            elif op.startswith("ds_read") or op.startswith("ds_load"): cls["ds_read"] += 1
            elif op.startswith("ds_"): cls["ds_write"] += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): cls["vmem"] += 1
            else: cls["other"] += 1
        print(f"  loop lines {a}-{b}: {len(seg)} instructions  " + "  ".join(f"{k} {v}" for k, v in cls.items()))
