import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantumattention_amd import _native
if os.environ.get("USE_DEV"): _native.LIB_PATH = _native.LIB_PATH.replace(".so", "_dev.so")
torch.manual_seed(0)
B, H, D, S = 1, 8, 128, 4096
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
q8, kf, vf, sq, sk, sv = _native.quant_qkv_fp8(q, k, v)
for f in (1.0, 0.3, 0.1, 0.01):
    for prec in ("fast", "accurate"):
        o = _native.fp8_attention_forward(q8, kf, vf, sq * f, sk, sv, Hkv=H, Skv=S, out_dtype=torch.bfloat16, is_causal=False, precision=prec)
        ref = torch.nn.functional.scaled_dot_product_attention(q.float() * f, k.float(), v.float())
        print("score scale", f, prec, "out vs sdpa(unquantised) max abs %.5f" % (o.float() - ref).abs().max().item())
