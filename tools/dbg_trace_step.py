"""Development: the fused step in a loop, for `rocprofv3 --kernel-trace --stats` (per-kernel durations of the pre-pass and the
attention launch inside the step).  QLIB=<path> selects another build of the library, PREC the precision, CAUSAL=1 the mask,
SHAPE=B,H,S,D the shape, MODE=fp8 (fp8_attn_func, default) | token (fp8_token_wise_attn_func) | bf16 (attn_func, the 16-bit path)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from quantumattention_amd import _native
if os.environ.get("QLIB"):
    _native.LIB_PATH = os.path.abspath(os.environ["QLIB"])
    _native.ABI_VERSION = int(os.environ.get("QLIB_ABI", _native.ABI_VERSION))
import quantumattention_amd as qa
B, H, S, D = (int(x) for x in os.environ.get("SHAPE", "4,32,4096,128").split(","))
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
fn = {"fp8": qa.fp8_attn_func, "token": qa.fp8_token_wise_attn_func, "bf16": qa.attn_func}[os.environ.get("MODE", "fp8")]
with qa.config.patch({"attention.precision": os.environ.get("PREC", "auto")}):
    for _ in range(int(os.environ.get("STEPS", "400"))): fn(q, k, v, is_causal=os.environ.get("CAUSAL") == "1")
torch.cuda.synchronize()
