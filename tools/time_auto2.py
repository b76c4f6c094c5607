"""Development: split the AUTO overhead of the causal D=128 kernel (dev library knobs QATTN_TWO_TERM_KEYS / QATTN_PEAK_R0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantumattention_amd import _native
_native.LIB_PATH = _native.LIB_PATH.replace(".so", "_dev.so")
B, H, S, D = 4, 32, 4096, 128
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
q8, kf, vf, sq, sk, sv = _native.quant_qkv_fp8(q, k, v)
def timeit(fn, n=30):
    for _ in range(60): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
out = []
for causal in (True, False):
    for prec in ("fast", "auto"):
        out.append("%s %s %.4f" % ("causal" if causal else "full", prec, timeit(lambda: _native.fp8_attention_forward(q8, kf, vf, sq, sk, sv, Hkv=H, Skv=S, out_dtype=torch.bfloat16, is_causal=causal, precision=prec))))
print({k_: v_ for k_, v_ in os.environ.items() if k_.startswith("QATTN_")}, " | ".join(out))
