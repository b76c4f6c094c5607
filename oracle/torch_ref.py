"""Torch-CPU restatement of the reference's *literal* eager op -- TEST INFRASTRUCTURE ONLY.

``fp8_attention_forward_ref`` follows src/quantum_attn/ops.py:64-95 step by step (cast q,k to the output
dtype, multiply by the scales cast to the output dtype, aten SDPA, contiguous).  It is pinned bit-for-bit to
the golden ``o1_*`` vectors (tests/test_oracle_golden.py) and is the ``cpu_baseline`` ("port") that bench.py
times on the host cores.  ``quantize_fp8_eager_ref`` follows src/quantum_attn/nn.py:14-19.
"""
from typing import Optional

import torch
import torch.nn.functional as F


def quantize_fp8_eager_ref(t: torch.Tensor, *, reduction_dim=-1, fp8_dtype=torch.float8_e4m3fn):
    # nn.py:14-19
    eps = torch.finfo(torch.float32).eps
    q_max = torch.finfo(fp8_dtype).max
    scale = t.abs().amax(reduction_dim, keepdim=True).mul(1.0 / q_max).clamp_min(eps)
    t_fp8 = (t / scale).clamp(-q_max, q_max).to(fp8_dtype)
    return t_fp8, scale.squeeze(reduction_dim).to(torch.float32)


def fp8_attention_forward_ref(query, key, value, scale_q: Optional[torch.Tensor] = None,
                              scale_k: Optional[torch.Tensor] = None, *, is_causal: bool = False,
                              scale: Optional[float] = None):
    # ops.py:64-95
    out_dtype = value.dtype
    query = query.to(out_dtype)
    key = key.to(out_dtype)
    if scale_q is not None:
        scale_q = scale_q.to(out_dtype)
        scale_k = scale_k.to(out_dtype)
        while scale_q.dim() < query.dim():
            scale_q = scale_q.unsqueeze(-1)
            scale_k = scale_k.unsqueeze(-1)
        query = query * scale_q
        key = key * scale_k
    return F.scaled_dot_product_attention(query, key, value, is_causal=is_causal, scale=scale).contiguous()


def attention_forward_ref(query, key, value, *, is_causal: bool = False, scale: Optional[float] = None):
    # ops.py:17-29
    return F.scaled_dot_product_attention(query, key, value, is_causal=is_causal, scale=scale).contiguous()
