#!/usr/bin/env python3
"""Development: the MFMA loops of the w4 kernel in a -save-temps .s file: instruction classes, accvgpr moves, scratch traffic.
   python tools/w4_loops.py <file.s> <kernel name substring> [--dump N]"""
import re, sys
from collections import Counter
f, pat = sys.argv[1], sys.argv[2]
dump = int(sys.argv[sys.argv.index("--dump") + 1]) if "--dump" in sys.argv else -1
text = open(f, errors="replace").read().split("\n")
st = [i for i, l in enumerate(text) if re.match(r"^_Z\w+:", l) and pat in l][0]
end = next(i for i in range(st + 1, len(text)) if ".end_amdhsa_kernel" in text[i] or re.match(r"^_Z\w+:", text[i]))
body = text[st:end]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
k = 0
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a = labels[m.group(1)]
        seg = [x.strip() for x in body[a:i + 1] if x.startswith("\t") and not x.strip().startswith((";", "."))]
        n = sum("v_mfma" in x for x in seg)
        if n < 8: continue
        c = Counter(x.split()[0] for x in seg)
        cls = Counter()
        for op, v in c.items():
            if "v_mfma" in op: cls["mfma"] += v
            elif "accvgpr" in op: cls["accvgpr"] += v
            elif op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): cls["lane"] += v
            elif op.startswith("v_"): cls["valu"] += v
            elif op == "s_waitcnt": cls["wait"] += v
            elif op == "s_nop": cls["nop"] += v
            elif op.startswith("s_"): cls["salu"] += v
            elif op.startswith("ds_"): cls["ds"] += v
            elif op.startswith("scratch"): cls["scratch"] += v
            else: cls["vmem"] += v
        print(k, "lines", a, i, "instr", len(seg), dict(cls))
        if k == dump:
            print("\n".join(body[a:i + 1]))
        k += 1
