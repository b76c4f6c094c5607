import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantumattention_amd import _native
_native.LIB_PATH = _native.LIB_PATH.replace(".so", "_dev.so")
import quantumattention_amd as qa
torch.manual_seed(0)
B, H, S, D = 1, 8, 4096, 128
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
q8, kf, vf, sq, sk, sv = _native.quant_qkv_fp8(q, k, v)
out, lse = _native.fp8_attention_forward(q8, kf, vf, sq, sk, sv, Hkv=H, Skv=S, out_dtype=torch.bfloat16, is_causal=False, return_lse=True, precision="fast")
qd = q8.float() * sq[..., None, None]
s = (qd[0, 0] @ (k[0, 0].float()).T) / D ** 0.5   # approx (k unquantised)
ref = torch.logsumexp(s, -1)
print("LSE kernel[:4]", lse[0, 0, :4].tolist(), "ref[:4]", ref[:4].tolist(), "max abs diff %.4f" % (lse[0, 0] - ref).abs().max().item())
d = (lse[0, 0] - ref)
print("diff by row%32:", [round(d[i::32].mean().item(), 3) for i in range(32)])
print("diff first rows:", [round(x, 3) for x in d[:40].tolist()])
for scale in (0.01,):
    out2, lse2 = _native.fp8_attention_forward(q8, kf, vf, sq * scale, sk, sv, Hkv=H, Skv=S, out_dtype=torch.bfloat16, is_causal=False, return_lse=True, precision="fast")
    ref2 = torch.logsumexp(s * scale, -1)
    print("flat scores: max abs diff %.4f" % (lse2[0, 0] - ref2).abs().max().item())
