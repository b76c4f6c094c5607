import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
B, H, S, D = 4, 32, 4096, 128
x = torch.randn(3 * B, H, S, D, dtype=torch.bfloat16, device="cuda")   # 403 MB
xi = x.view(torch.int16)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
nb = x.numel() * 2
for name, fn in (("int16 max", lambda: xi.max()), ("bf16 abs().amax(dims 2,3)", lambda: x.abs().amax((2, 3))), ("bf16 sum", lambda: x.sum()),
                 ("clone", lambda: x.clone())):
    t = timeit(fn)
    print(f"{name:28s} {t:.4f} ms  {nb / t / 1e9:.2f} TB/s (read side)")
