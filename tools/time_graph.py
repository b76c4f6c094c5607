"""Development: the fused step eager vs replayed from a HIP graph, steady state."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantumattention_amd as qa
B, H, S, D = 4, 32, 4096, 128
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
prec = os.environ.get("PREC", "auto")
def timeit(fn, n=300):
    for _ in range(300): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with qa.config.patch({"attention.precision": prec}):
    step = lambda: qa.fp8_attn_func(q, k, v)
    eager = timeit(step)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    graph = timeit(g.replay)
    g4 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g4):
        for _ in range(4): out = step()
    graph4 = timeit(g4.replay, 100) / 4
print("prec", prec, "eager %.4f ms  graph(1 step) %.4f ms  graph(4 steps) %.4f ms/step" % (eager, graph, graph4))
