#!/usr/bin/env python3
"""Development: does tracing the abs-max / sum-of-squares reductions into the caller's graph pay when the producer of q, k, v is an
EXTERN GEMM (qkv = linear(x): nothing for Inductor to fuse them into)?  Times projection + fp8_attn_func three ways on one box, interleaved:
eager; torch.compile with the reductions inlined (config.attention.inline_abs_max_under_compile, the default); torch.compile without.
  python tools/time_compiled_producer.py [B H S D]"""
import os, sys, statistics
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantumattention_amd as qa  # noqa: E402

B, H, S, D = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (4, 32, 4096, 128)
torch.manual_seed(0)
E = H * D
x = torch.randn(B, S, E, device="cuda", dtype=torch.bfloat16)
w = [torch.randn(E, E, device="cuda", dtype=torch.bfloat16) / E ** 0.5 for _ in range(3)]


def block(x):
    q, k, v = ((x @ wi).view(B, S, H, D).transpose(1, 2).contiguous() for wi in w)
    return qa.fp8_attn_func(q, k, v)


def proj_only(x):
    return [(x @ wi).view(B, S, H, D).transpose(1, 2).contiguous() for wi in w]


def timed(fn, n=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn(x)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


variants = {"eager": block}
with qa.config.patch({"attention.inline_abs_max_under_compile": True}):
    c_in = torch.compile(block, backend="inductor"); c_in(x)
variants["compiled, reductions inlined"] = c_in
torch._dynamo.reset()
with qa.config.patch({"attention.inline_abs_max_under_compile": False}):
    c_out = torch.compile(block, backend="inductor"); c_out(x)
    variants["compiled, abs-max pass in the op"] = c_out
    variants["projection alone (eager)"] = proj_only
    for fn in variants.values():
        for _ in range(5):
            fn(x)
    res = {k: [] for k in variants}
    for r in range(7):
        for k_, fn in variants.items():
            res[k_].append(timed(fn))
print(f"B{B} H{H} S{S} D{D}: x[{B},{S},{E}] @ 3 x W[{E},{E}] -> q, k, v -> fp8_attn_func (ms, median of 7 x 10, interleaved)")
for k_, v_ in res.items():
    print(f"  {k_:36s} {statistics.median(v_):.4f}")
