import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantumattention_amd import _native
_native.LIB_PATH = _native.LIB_PATH.replace(".so", "_dev.so")
import quantumattention_amd as qa
torch.manual_seed(0)
q, k, v = (torch.randn(1, 8, 4096, 128, dtype=torch.bfloat16, device="cuda") for _ in range(3))
with qa.config.patch({"attention.precision": "auto"}):
    a = qa.fp8_attn_func(q, k, v)
torch.cuda.synchronize()
s = (q[0, 0].float() @ k[0, 0].float().T) / 128 ** 0.5
w = torch.softmax(s, -1)
R = 1 / w.max(-1).values
print("true R head0: min %.2f p1 %.2f med %.2f | row0 %.2f row777 %.2f | score std %.3f rowmax mean %.3f" % (R.min(), R.kthvalue(41).values, R.median(), R[0], R[777], s.std(), s.max(-1).values.mean()))
l = torch.exp(s - s.max(-1, keepdim=True).values).sum(-1) * 32
print("expected l' (delta=0) row0 %.1f row777 %.1f" % (l[0], l[777]))
