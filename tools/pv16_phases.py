"""Development: per-phase cycles of the two-group 16-bit-V loop (library built with -DQATTN_PV16_STAMP)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quantumattention_amd import _native
torch.manual_seed(0)
B, H, S, D = 4, 32, 4096, 128
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
q8, sq = _native.quant_fp8(q); kf, sk = _native.quant_fp8(k, layout=_native.LAYOUT_KFRAG)
for _ in range(3):
    out, lse = _native.fp8_attention_forward(q8, kf, v, sq, sk, None, Hkv=H, Skv=S, out_dtype=torch.bfloat16, is_causal=False, return_lse=True)
torch.cuda.synchronize()
l = lse.float().cpu().numpy().reshape(B * H, S // 32, 32)[:, :, 1:6]     # per wave: products, softmax, mid barrier, tail barrier, dma issue
grp = (np.arange(S // 32) % 8) // 4
names = ["products", "softmax", "mid barrier", "tail barrier", "dma issue"]
for g in (0, 1):
    x = l[:, grp == g].reshape(-1, 5)
    print(f"group {g}: " + " | ".join(f"{n} {np.median(x[:, i]) / 65:.0f}" for i, n in enumerate(names)) + f" | total {np.median(x.sum(1)) / 65:.0f}  (cycles per trip, 65 trips)")
