#!/usr/bin/env python3
"""Soak: the dynamic block hand-out (causal launches, large non-causal ones) run back to back for SECONDS per shape, every
result compared bit for bit with the first -- a block handed out twice or not at all, a race in the relaxed counters or a rescue
that depends on who computes it would show up as a differing output sooner or later.   python tools/soak.py [seconds per shape = 20]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantumattention_amd as qa  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
SHAPES = [(4, 32, 4096, 128, True, "auto", 1.0), (16, 16, 2048, 128, True, "auto", 1.3), (8, 32, 6144, 128, False, "auto", 1.0),
          (4, 40, 16384, 128, True, "fast", 1.0), (3, 5, 3000, 128, True, "auto", 1.0), (4, 32, 4096, 64, True, "auto", 1.0)]
bad = 0
for B, H, S, D, causal, prec, spread in SHAPES:
    torch.manual_seed(S + D)
    q = (torch.randn(B, H, S, D, device="cuda") * spread).to(torch.bfloat16)
    k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(2))
    with qa.config.patch({"attention.precision": prec}):
        first = qa.fp8_attn_func(q, k, v, is_causal=causal).clone()
        n = diff = 0
        t0 = time.time()
        while time.time() - t0 < secs:
            outs = [qa.fp8_attn_func(q, k, v, is_causal=causal) for _ in range(8)]
            torch.cuda.synchronize()
            for o in outs:
                n += 1
                diff += not torch.equal(o, first)
    bad += diff
    print(f"{'ok  ' if not diff else 'FAIL'} B{B} H{H} S{S} D{D} {'causal' if causal else 'full'} {prec} q x{spread}: {n} launches in {secs:.0f} s, {diff} differ from the first", flush=True)
sys.exit(1 if bad else 0)
