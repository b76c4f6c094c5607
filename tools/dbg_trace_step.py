import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import quantumattention_amd as qa
B, H, S, D = 4, 32, 4096, 128
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
with qa.config.patch({"attention.precision": "fast"}):
    for _ in range(300): qa.fp8_attn_func(q, k, v)
torch.cuda.synchronize()
