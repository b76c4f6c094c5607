#!/usr/bin/env python3
"""Development: do two builds of the library produce the same BITS?  (for changes that must not alter results: scheduling, block
boundaries, prefetches).  Loads both through ctypes (tools/ab.py's Variant), runs the fused step and the separate attention call on
the same inputs for a few shapes / precisions, prints equal / max-abs difference per case; exit status 1 if any case differs.
   python tools/cmp_libs.py new=quantumattention_amd/libqattn_hip.so old=tools/ab_libs/libqattn_r3.so"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ab  # noqa: E402

libs = [a.split("=", 1) for a in sys.argv[1:] if "=" in a]
assert len(libs) == 2, __doc__
skip_fused_causal = "--attn-only" in sys.argv
bad = 0
for (B, H, S, D), causal, scale in [((4, 32, 4096, 128), False, 1.0), ((2, 8, 4096, 128), True, 1.0), ((1, 8, 2304, 128), False, 2.0),
                                     ((2, 5, 3000, 128), True, 1.3), ((1, 40, 16384, 128), True, 1.0)]:
    torch.manual_seed(S + B)
    q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    q = (q.float() * scale).to(torch.bfloat16)
    vs = [ab.Variant(n, ab.load(p), q, k, v, causal, 0) for n, p in libs]
    for path in ("fused", "attn"):
        for prec in ("auto", "fast", "accurate"):
            outs = []
            for x in vs:
                x.out.fill_(float("nan"))
                getattr(x, path)(ab.PREC[prec])
                torch.cuda.synchronize()
                outs.append(x.out.clone())
            same = torch.equal(outs[0], outs[1])
            diff = (outs[0].float() - outs[1].float()).abs().max().item()
            rows = ((outs[0] != outs[1]).any(dim=-1)).sum().item()
            bad += not same
            print(f"B{B} H{H} S{S} {'causal' if causal else 'full  '} q x{scale} {path:5s} {prec:8s}: {'same bits' if same else f'DIFFERENT: max-abs {diff:.5f}, {rows} rows differ'}", flush=True)
    del vs
sys.exit(1 if bad else 0)
