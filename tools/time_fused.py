"""Development: the fused step and its in-step attention time (library events), dev library; QATTN_NO_VBLOCK=1 = one V scale per head."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantumattention_amd import _native
_native.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab_libs", "libqattn_dev.so")
import quantumattention_amd as qa
B, H, S, D = 4, 32, 4096, 128
causal = "--causal" in sys.argv
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
L = _native.lib()
out = []
for prec in ("fast", "auto"):
    with qa.config.patch({"attention.precision": prec}):
        fn = lambda: qa.fp8_attn_func(q, k, v, is_causal=causal)
        for _ in range(300): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): fn()
        e1.record(); torch.cuda.synchronize()
        step = e0.elapsed_time(e1) / 200
        L.qattn_profile_attention(1)
        tot = 0.0
        for _ in range(5):
            for _ in range(20): fn()
            tot += L.qattn_last_attention_ms()
        L.qattn_profile_attention(0)
        out.append("%s step %.4f attn-in-step %.4f" % (prec, step, tot / 5))
print({k_: v_ for k_, v_ in os.environ.items() if k_.startswith("QATTN_")}, " | ".join(out))
