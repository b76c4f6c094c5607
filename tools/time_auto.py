"""Development: attention-only time of the D=128 head-wise kernel, auto vs fast, flat and peaked inputs (product library; QLIB=<path>: another build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantumattention_amd import _native
if os.environ.get("QLIB"): _native.LIB_PATH = os.path.abspath(os.environ["QLIB"])
B, H, S, D = 4, 32, 4096, 128
torch.manual_seed(0)
def timeit(fn, n=30):
    for _ in range(60): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
res = []
for qs in (1.0, 2.0, 3.0):
    q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    q = q * qs
    q8, kf, vf, sq, sk, sv = _native.quant_qkv_fp8(q, k, v)
    for causal in (False, True):
        row = []
        for prec in ("fast", "auto"):
            row.append(timeit(lambda: _native.fp8_attention_forward(q8, kf, vf, sq, sk, sv, Hkv=H, Skv=S, out_dtype=torch.bfloat16, is_causal=causal, precision=prec)))
        res.append("qx%g %s fast %.4f auto %.4f (+%.1f%%)" % (qs, "causal" if causal else "full", row[0], row[1], 100 * (row[1] / row[0] - 1)))
print({k_: v_ for k_, v_ in os.environ.items() if k_.startswith("QATTN_")}, " | ".join(res))
# fused step (pre-pass + attention): the AUTO predictor sees the heads' sums of squares here
import quantumattention_amd as qa
res = []
for qs in (1.0, 1.25, 1.5, 2.0, 3.0):
    q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    q = q * qs
    for causal in (False, True):
        row = []
        for prec in ("fast", "auto", "accurate"):
            with qa.config.patch({"attention.precision": prec}):
                row.append(timeit(lambda: qa.fp8_attn_func(q, k, v, is_causal=causal)))
        res.append("qx%g %s fast %.4f auto %.4f accurate %.4f" % (qs, "causal" if causal else "full", *row))
print("fused step:", " | ".join(res))
