"""Time the quant pre-pass variants at the BASELINE config-2 shape (development aid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantumattention_amd import _native

B, H, S, D = 4, 32, 4096, 128
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))


def timeit(fn, n=30):
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


def three():
    _native.quant_fp8(q, layout=_native.LAYOUT_ROWMAJOR)
    _native.quant_fp8(k, layout=_native.LAYOUT_KFRAG)
    _native.quant_fp8(v, layout=_native.LAYOUT_VFRAG)


print("fused qkv      %.4f ms" % timeit(lambda: _native.quant_qkv_fp8(q, k, v)))
print("three tensors  %.4f ms" % timeit(three))
print("q only         %.4f ms" % timeit(lambda: _native.quant_fp8(q, layout=_native.LAYOUT_ROWMAJOR)))
print("k only (KFRAG) %.4f ms" % timeit(lambda: _native.quant_fp8(k, layout=_native.LAYOUT_KFRAG)))
print("v only (VFRAG) %.4f ms" % timeit(lambda: _native.quant_fp8(v, layout=_native.LAYOUT_VFRAG)))
print("copy 134MB->134MB  %.4f ms" % timeit(lambda: q.clone()))
