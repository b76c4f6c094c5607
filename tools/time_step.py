"""Development: time the fused step and the attention kernel alone (product library; QLIB=<path>: another build, e.g. a tools/ab_libs variant)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantumattention_amd import _native
if os.environ.get("QLIB"): _native.LIB_PATH = os.path.abspath(os.environ["QLIB"])
import quantumattention_amd as qa
B, H, S, D = 4, 32, 4096, 128
causal = "--causal" in sys.argv
prec = os.environ.get("PREC", "auto")
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
def timeit(fn, n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with qa.config.patch({"attention.precision": prec}):
    step = timeit(lambda: qa.fp8_attn_func(q, k, v, is_causal=causal)) if os.environ.get("ONLY", "") != "attn" else 0.0
    q8, kf, vf, sq, sk, sv = _native.quant_qkv_fp8(q, k, v)
    attn = timeit(lambda: _native.fp8_attention_forward(q8, kf, vf, sq, sk, sv, Hkv=H, Skv=S, out_dtype=torch.bfloat16, is_causal=causal, precision=prec))
    quant = timeit(lambda: _native.quant_qkv_fp8(q, k, v))
print("env", {k_: v_ for k_, v_ in os.environ.items() if k_.startswith("QATTN_") or k_ == "QLIB"}, "prec", prec, "| step %.4f ms  attn-only %.4f ms  quant %.4f ms" % (step, attn, quant))
