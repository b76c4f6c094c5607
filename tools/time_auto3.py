"""Development: fused-step time of AUTO at several score spreads, with / without the mid-sweep forecast (dev library, QATTN_NO_FORECAST=1 = off)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantumattention_amd import _native
_native.LIB_PATH = _native.LIB_PATH.replace(".so", "_dev.so")
import quantumattention_amd as qa
B, H, S, D = 4, 32, 4096, 128
torch.manual_seed(0)
def timeit(fn, n=20):
    for _ in range(40): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
out = []
for qs in (1.0, 1.15, 1.25, 1.4, 2.0):
    q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    q = q * qs
    with qa.config.patch({"attention.precision": "auto"}):
        out.append("qx%g %.4f" % (qs, timeit(lambda: qa.fp8_attn_func(q, k, v))))
# anisotropic: two dimensions of q and k carry 3x the amplitude (isotropic moment estimate 1.27, true score variance 2.25)
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
q[..., :2] *= 3; k[..., :2] *= 3
with qa.config.patch({"attention.precision": "auto"}):
    out.append("aniso %.4f" % timeit(lambda: qa.fp8_attn_func(q, k, v)))
with qa.config.patch({"attention.precision": "accurate"}):
    out.append("aniso-accurate %.4f" % timeit(lambda: qa.fp8_attn_func(q, k, v)))
print({k_: v_ for k_, v_ in os.environ.items() if k_.startswith("QATTN_")}, " | ".join(out))
