"""A few launches of one configuration for rocprofv3 --kernel-trace --stats: SHAPE=B,H,S,D CAUSAL=0|1 PREC=auto|fast|accurate MODE=fused|attn|v16|bf16 SCALING=head-wise|token-wise STEPS=n VIEWS=0|1"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantumattention_amd as qa
from quantumattention_amd import _native
B, H, S, D = (int(x) for x in os.environ.get("SHAPE", "4,32,4096,128").split(","))
causal = os.environ.get("CAUSAL", "0") == "1"
prec, mode, scaling = os.environ.get("PREC", "auto"), os.environ.get("MODE", "fused"), os.environ.get("SCALING", "head-wise")
steps = int(os.environ.get("STEPS", "20"))
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
if os.environ.get("VIEWS", "0") == "1":   # the same values as transposed views of [B,S,H,D] tensors (include/qattn_strided.h)
    q, k, v = (t.transpose(1, 2).contiguous().transpose(1, 2) for t in (q, k, v))
if mode == "bf16":   # the 16-bit sibling path (attn_func: qattn_attention_forward_16 behind quantum_attn::attention_forward)
    for _ in range(steps + 3): qa.attn_func(q, k, v, is_causal=causal)
elif mode == "fused":
    fn = qa.fp8_attn_func if scaling == "head-wise" else qa.fp8_token_wise_attn_func
    with qa.config.patch({"attention.precision": prec}):
        for _ in range(steps + 3): fn(q, k, v, is_causal=causal)
else:
    q8, sq = _native.quant_fp8(q, scaling=scaling)
    kf, sk = _native.quant_fp8(k, scaling=scaling, layout=_native.LAYOUT_KFRAG)
    vf, sv = _native.quant_fp8(v, layout=_native.LAYOUT_VFRAG)
    for _ in range(steps + 3):
        if mode == "v16": _native.fp8_attention_forward(q8, kf, v, sq, sk, None, Hkv=H, Skv=S, out_dtype=torch.bfloat16, is_causal=causal, scaling=scaling)
        else: _native.fp8_attention_forward(q8, kf, vf, sq, sk, sv, Hkv=H, Skv=S, out_dtype=torch.bfloat16, is_causal=causal, scaling=scaling, precision=prec)
torch.cuda.synchronize()
