import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantumattention_amd.utils.shard import synthetic_qkv
H, S, D = 32, 4096, 128
def stats(name, q, k):
    fr = []
    for h in range(q.shape[1]):
        s = (q[0, h].float() @ k[0, h].float().T) / D ** 0.5
        w = torch.softmax(s, -1).max(-1).values
        fr.append(((1 / w) < 24).float().mean().item())
    q32 = q.float()
    print(name, "P(row R<24) = %.2e" % (sum(fr) / len(fr)), "| q std %.4f kurtosis %.3f max|q| %.2f" % (q32.std(), ((q32 - q32.mean()) ** 4).mean() / q32.var() ** 2, q32.abs().max()))
torch.manual_seed(0)
q, k, v = (torch.randn(1, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
stats("cuda randn bf16      ", q, k)
q, k, v = synthetic_qkv(range(1), H, S, D, device="cuda")
stats("cpu generator -> bf16", q, k)
q, k = (torch.randn(1, H, S, D, device="cuda").to(torch.bfloat16) for _ in range(2))
stats("cuda randn f32->bf16 ", q, k)
