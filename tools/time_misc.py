"""Time the side paths at the BASELINE shapes: 16-bit exact / fast, token-wise (templated kernel), D = 64 / 256 (product library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantumattention_amd as qa
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
B, H, S = 4, 32, 4096
for D in (128, 64, 256):
    q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    fl = 4.0 * B * H * S * S * D
    if D == 128:
        kf, vf = _native.pack16(k, _native.LAYOUT_K16FRAG), _native.pack16(v, _native.LAYOUT_V16FRAG)
        for fast in (False, True):
            t = timeit(lambda: _native.attention_forward_16(q, kf, vf, Hkv=H, Skv=S, is_causal=False, fast_exp=fast))
            print(f"16-bit D=128 {'fast' if fast else 'exact'} exp: {t:.4f} ms = {fl / t / 1e9:.0f} TF")
    for scaling in (("head-wise", "token-wise") if D == 128 else ("head-wise",)):
        for causal in (False, True):
            for prec in ("fast", "auto"):
                q8, sq = _native.quant_fp8(q, scaling=scaling)
                kfr, sk = _native.quant_fp8(k, scaling=scaling, layout=_native.LAYOUT_KFRAG)
                vfr, sv = _native.quant_fp8(v, layout=_native.LAYOUT_VFRAG)
                t = timeit(lambda: _native.fp8_attention_forward(q8, kfr, vfr, sq, sk, sv, Hkv=H, Skv=S, out_dtype=torch.bfloat16, is_causal=causal, scaling=scaling, precision=prec))
                f = fl / 2 if causal else fl
                print(f"fp8 D={D} {scaling} {'causal' if causal else 'full  '} {prec}: {t:.4f} ms = {f / t / 1e9:.0f} TF")
