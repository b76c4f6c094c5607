#!/usr/bin/env python3
"""Model behind kCrushMean (qattn_attn.h row_is_peaked): rows whose top key is the exact reference and 8 .. 10 nats above an
otherwise flat row.  The exact-top rule judged such a row by its rest alone; the rest's P' then sits at or below e4m3's smallest
normal (2^-6 against the reference 2^5) and is crushed.  Bins the rows the rule accepted by the mean P' of their other keys and
prints the worst one-term error per bin (tools/models/sim_exact_top_n.py arithmetic; found by tools/fuzz_parity.py).
   python tools/models/sim_crushed_rest.py"""
import math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # the repository root
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sim_exact_top_n import sim, rules
from sim_exact_top import ref_out

D = 128; c0 = 1.4426950408889634 / math.sqrt(D)
bins = [(0.0, 0.02), (0.02, 0.03), (0.03, 0.04), (0.04, 0.05), (0.05, 0.0625), (0.0625, 0.08), (0.08, 0.12), (0.12, 9.0)]
errs = {b: [] for b in bins}
for seed in range(10):
    for n, mult in ((1760, 3.0), (4096, 3.0), (2048, 2.5), (8192, 3.0) if seed < 3 else (1024, 3.0)):
        g = torch.Generator().manual_seed(100 + seed)
        q, k, v = (torch.randn(n, D, generator=g) for _ in range(3))
        k[n // 3] *= mult                       # one key (two on odd seeds) with a multiple of the others' norm
        if seed % 2: k[n // 2] *= mult * 0.8
        out, l, q2, ptop, exact = sim(q, k, v, c0, False)
        err = np.abs(out - ref_out(q, k, v, c0, False)).max(1)
        _, flagged = rules(l, q2, ptop, exact, np.full(n, float(n)))     # the rule WITHOUT the dynamic-range floor
        sel = ~flagged & exact & (l / ptop < 24)
        mean_rest = (l - 32.0) / n
        for b in bins:
            errs[b] += list(err[sel & (mean_rest >= b[0]) & (mean_rest < b[1])])
for b in bins:
    e = np.array(errs[b])
    print(f"mean rest P' in [{b[0]}, {b[1]}): {len(e):4d} accepted rows, worst one-term error {e.max() if len(e) else 0:.4f}, p99 {np.percentile(e, 99) if len(e) else 0:.4f}")
