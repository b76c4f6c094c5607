"""Time the 16-bit-V mode of the fp8 entry (csrc/qattn_pv16.h) beside the fp8-V path's three precisions, attention launch only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantumattention_amd import _native

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

torch.manual_seed(0)
shapes = [(4, 32, 4096, 128), (4, 32, 4096, 64), (4, 32, 4096, 256), (1, 32, 16384, 128)]
for (B, H, S, D) in shapes:
    q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    q8, sq = _native.quant_fp8(q)
    kf, sk = _native.quant_fp8(k, layout=_native.LAYOUT_KFRAG)
    vf, sv = _native.quant_fp8(v, layout=_native.LAYOUT_VFRAG)
    for causal in (False, True):
        fl = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
        row = []
        for prec in ("fast", "auto", "accurate"):
            t = timeit(lambda: _native.fp8_attention_forward(q8, kf, vf, sq, sk, sv, Hkv=H, Skv=S, out_dtype=torch.bfloat16, is_causal=causal, precision=prec))
            row.append(f"{prec} {t:.4f} ms ({fl / t / 1e9:.0f} TF)")
        t = timeit(lambda: _native.fp8_attention_forward(q8, kf, v, sq, sk, None, Hkv=H, Skv=S, out_dtype=torch.bfloat16, is_causal=causal))
        row.append(f"16-bit V {t:.4f} ms ({fl / t / 1e9:.0f} TF)")
        print(f"B{B} H{H} S{S} D{D} {'causal' if causal else 'full  '}: " + " | ".join(row), flush=True)
