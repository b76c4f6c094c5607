"""Development: time the q/k/v pre-pass alone (dev library; QATTN_ONE_READ / QATTN_SLICE_KB switches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantumattention_amd import _native
if os.environ.get("USE_DEV", "1") == "1": _native.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab_libs", "libqattn_dev.so")
B, H, S, D = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (4, 32, 4096, 128)))
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
def timeit(fn, n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t = timeit(lambda: _native.quant_qkv_fp8(q, k, v))
print({k_: v_ for k_, v_ in os.environ.items() if k_.startswith("QATTN_")}, (B, H, S, D), "quant_qkv %.4f ms" % t)
