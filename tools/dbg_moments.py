"""Development: read back the pre-pass's sums of squares from the fused step's workspace and compare with torch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantumattention_amd import _native
from quantumattention_amd._native import *
B, H, S, D = 2, 4, 1000, 128
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
q = q * float(os.environ.get("QS", "1"))
L = _native.lib()
dev = q.device
out = torch.empty_like(q)
q8 = torch.empty((B, H, S, D), dtype=torch.uint8, device=dev)
kf = torch.empty((L.qattn_fp8_tensor_bytes(LAYOUT_KFRAG, B, H, S, D),), dtype=torch.uint8, device=dev)
vf = torch.empty((L.qattn_fp8_tensor_bytes(LAYOUT_VFRAG, B, H, S, D),), dtype=torch.uint8, device=dev)
sq = torch.empty((B, H), dtype=torch.float32, device=dev); sk = torch.empty_like(sq); sv = torch.empty_like(sq)
ws_bytes = L.qattn_fp8_quant_attention_workspace_bytes(B, H, H, S)
ws = torch.full((ws_bytes,), 0xAB, dtype=torch.uint8, device=dev)
rc = L.qattn_fp8_quant_attention_forward(q.data_ptr(), k.data_ptr(), v.data_ptr(), fmt_of(q.dtype), out.data_ptr(), q8.data_ptr(), kf.data_ptr(),
    vf.data_ptr(), sq.data_ptr(), sk.data_ptr(), sv.data_ptr(), B, H, H, S, S, D, 0, SCALE_HEAD, 0, 0, 0.0, PRECISION["auto"], ws.data_ptr(), ws_bytes, _native._stream(q))
torch.cuda.synchronize()
print("rc", rc)
n = B * H
w = ws.view(torch.int32)
K = 256
ssq = ws.view(torch.float32)[K * 3 * n: K * 5 * n].view(2 * n, K)[:, :8].sum(1)
print("amax bits q", w[:K * n].view(n, K)[:, :8].max(1).values.tolist()[:4])
print("ssq q", ssq[:n].tolist()); print("ref  ", q.float().pow(2).sum((2, 3)).flatten().tolist())
print("ssq k", ssq[n:].tolist()); print("ref  ", k.float().pow(2).sum((2, 3)).flatten().tolist())
ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float())
print("max err vs fp32 sdpa on unquantised", (out.float() - ref).abs().max().item())
