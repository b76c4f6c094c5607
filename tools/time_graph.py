"""Whole step (quant pre-pass + attention) eager vs replayed from a HIP graph, BASELINE config-2 shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantumattention_amd as qa

B, H, S, D = 4, 32, 4096, 128
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
for _ in range(5):
    qa.fp8_attn_func(q, k, v)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = qa.fp8_attn_func(q, k, v)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print("eager step  %.4f ms" % timeit(lambda: qa.fp8_attn_func(q, k, v)))
print("graph step  %.4f ms" % timeit(g.replay))
