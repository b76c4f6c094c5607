import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantumattention_amd import _native
_native.LIB_PATH = _native.LIB_PATH.replace(".so", "_dev.so")
torch.manual_seed(0)
B, H, S, D = 1, 8, 4096, 128
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
q8, kf, vf, sq, sk, sv = _native.quant_qkv_fp8(q, k, v)
k8, _ = _native.quant_fp8(k)
out = _native.fp8_attention_forward(q8, kf, vf, sq, sk, sv, Hkv=H, Skv=S, out_dtype=torch.bfloat16, is_causal=False, precision="auto")
torch.cuda.synchronize()
n = 0
for h in range(H):
    s = ((q8[0, h].float() * sq[0, h]) @ (k8[0, h].float() * sk[0, h]).T) / D ** 0.5
    w = torch.softmax(s, -1).max(-1).values
    R = 1 / w
    bad = (R < 24).nonzero().flatten().tolist()
    n += len(bad)
    for i in bad[:3]: print("true: head", h, "row", i, "R %.2f" % R[i].item())
print("true rows with R<24:", n, "of", H * S)
