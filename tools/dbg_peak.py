"""Development: fraction of rows the auto-precision redo changes, as a function of QATTN_PEAK_R0 (dev library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantumattention_amd import _native
_native.LIB_PATH = _native.LIB_PATH.replace(".so", "_dev.so")
import quantumattention_amd as qa

torch.manual_seed(0)
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
q, k, v = (torch.randn(2, 8, 4096, 128, dtype=torch.bfloat16, device="cuda") for _ in range(3))
q = q * scale
with qa.config.patch({"attention.precision": "fast"}):
    f = qa.fp8_attn_func(q, k, v)
with qa.config.patch({"attention.precision": "auto"}):
    a = qa.fp8_attn_func(q, k, v)
with qa.config.patch({"attention.precision": "accurate"}):
    c = qa.fp8_attn_func(q, k, v)
ch = (a != f).any(-1).float()
print("R0", os.environ.get("QATTN_PEAK_R0"), "scale", scale, "changed rows %.4f" % ch.mean().item(), "blocks changed %.4f" % (ch.view(2, 8, 16, 256).mean(-1) > 0.5).float().mean().item(),
      "| auto==accurate rows %.4f" % (a == c).all(-1).float().mean().item())
