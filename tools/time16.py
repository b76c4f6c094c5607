"""Time the 16-bit kernel (attention only, K/V pre-packed) and the whole attn_func step at a BASELINE shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantumattention_amd as qa
from quantumattention_amd import _native

B, H, S, D = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (4, 32, 4096, 128)))
causal = len(sys.argv) > 5 and sys.argv[5] == "causal"
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
kf = _native.pack16(k, _native.LAYOUT_K16FRAG)
vf = _native.pack16(v, _native.LAYOUT_V16FRAG)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


flops = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
t_k = timeit(lambda: _native.attention_forward_16(q, kf, vf, Hkv=H, Skv=S, is_causal=causal))
t_s = timeit(lambda: qa.attn_func(q, k, v, is_causal=causal))
t_ref = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=causal), n=5)
print(f"16-bit kernel {t_k:.4f} ms = {flops / t_k / 1e9:.1f} TFLOP/s | attn_func step {t_s:.4f} ms = {flops / t_s / 1e9:.1f} "
      f"TFLOP/s | aten SDPA {t_ref:.4f} ms = {flops / t_ref / 1e9:.1f} TFLOP/s")
