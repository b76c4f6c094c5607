import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantumattention_amd import _native
_native.LIB_PATH = _native.LIB_PATH.replace(".so", "_dev.so")
torch.manual_seed(0)
B, H, D, S = 1, 8, 128, 4096
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
q8, kf, vf, sq, sk, sv = _native.quant_qkv_fp8(q, k, v)
o_byte_lse, lse = _native.fp8_attention_forward(q8, kf, vf, sq * 0.01, sk, sv, Hkv=H, Skv=S, out_dtype=torch.bfloat16, is_causal=False, return_lse=True, precision="fast")
ref = torch.nn.functional.scaled_dot_product_attention(q.float() * 0.01, k.float(), v.float())
print("flat: out vs sdpa max abs %.5f | LSE - ln(S) first rows" % (o_byte_lse.float() - ref).abs().max().item(), [round(x - 8.3178, 3) for x in lse[0, 0, :16].tolist()])
