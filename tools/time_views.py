#!/usr/bin/env python3
"""Timing on the GPU box: the fused step on dense [B,H,S,D] tensors, on transposed views of [B,S,H,D] tensors (include/qattn_strided.h), and on
the same views behind three .contiguous() copies (what a dense-only entry costs such a caller) -- C2 / C3 shapes, token-wise, D = 64 / 256.
   python tools/time_views.py [calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantumattention_amd as qa

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def t_ms(fn, n=calls, settle=0.3):
    t_end = time.perf_counter() + settle
    while time.perf_counter() < t_end:
        for _ in range(5): fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = []
    for _ in range(3):
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / n)
    return sorted(best)[1]


print(f"{'config':44s} {'dense':>8s} {'views':>8s} {'copies':>8s}  views/dense  copies/views")
for name, B, H, S, D, causal, tok in [("C2  B4 H32 S4096 D128 full head-wise", 4, 32, 4096, 128, False, False),
                                      ("C3  B4 H32 S4096 D128 causal head-wise", 4, 32, 4096, 128, True, False),
                                      ("    B4 H32 S4096 D128 full token-wise", 4, 32, 4096, 128, False, True),
                                      ("    B4 H32 S4096 D128 causal token-wise", 4, 32, 4096, 128, True, True),
                                      ("    B4 H32 S4096 D64 causal head-wise", 4, 32, 4096, 64, True, False),
                                      ("    B4 H32 S4096 D256 full head-wise", 4, 32, 4096, 256, False, False),
                                      ("    B2 H16 S16384 D128 causal head-wise", 2, 16, 16384, 128, True, False)]:
    xs = [torch.randn(B, S, H, D, dtype=torch.bfloat16, device="cuda") for _ in range(3)]
    qv, kv, vv = (x.transpose(1, 2) for x in xs)
    qd, kd, vd = (t.contiguous() for t in (qv, kv, vv))
    fn = qa.fp8_token_wise_attn_func if tok else qa.fp8_attn_func
    assert torch.equal(fn(qv, kv, vv, is_causal=causal), fn(qd, kd, vd, is_causal=causal))
    d = t_ms(lambda: fn(qd, kd, vd, is_causal=causal))
    v = t_ms(lambda: fn(qv, kv, vv, is_causal=causal))
    c = t_ms(lambda: fn(qv.contiguous(), kv.contiguous(), vv.contiguous(), is_causal=causal))
    print(f"{name:44s} {d:8.4f} {v:8.4f} {c:8.4f}  {v / d:11.3f}  {c / v:12.3f}")
    del xs, qv, kv, vv, qd, kd, vd
