import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantumattention_amd import _native
_native.LIB_PATH = _native.LIB_PATH.replace(".so", "_dev.so")
torch.manual_seed(0)
B, H, D = 1, 8, 128
for S in (64, 128, 192, 256, 512, 1024, 4096):
    q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    q8, kf, vf, sq, sk, sv = _native.quant_qkv_fp8(q, k, v)
    out, lse = _native.fp8_attention_forward(q8, kf, vf, sq, sk, sv, Hkv=H, Skv=S, out_dtype=torch.bfloat16, is_causal=False, return_lse=True, precision="fast")
    qd = q8.float() * sq[..., None, None]
    s = (qd[0, 0] @ (k[0, 0].float()).T) / D ** 0.5
    ref = torch.logsumexp(s, -1)
    d = lse[0, 0] - ref
    print("S", S, "LSE diff: max abs %.4f mean %.4f | first rows" % (d.abs().max().item(), d.mean().item()), [round(x, 3) for x in d[:8].tolist()])
