"""User call surface.  Mirrors src/quantum_attn/quantum_attn_interface.py: SDPA-signature functions and the three
`*_with_fallback` ops (CompositeImplicitAutograd, interface.py:21-35) that fall back to
torch.nn.functional.scaled_dot_product_attention when the input is unsupported (interface.py:62-98, 130-176,
205-248).  The reference's latent TypeError in fp8_token_wise_attn_func_with_fallback (it forwards a
`scaling_method=` kwarg the callee does not accept, interface.py:229-238 vs :179-190) is not inherited."""
from typing import Optional

import torch
import torch.nn.functional as F

from .nn import attention, can_use_attention, dynamically_quantize_fp8, fp8_attention

__all__ = [
    "attn_func",
    "attn_func_with_fallback",
    "fp8_attn_func",
    "fp8_attn_func_with_fallback",
    "fp8_token_wise_attn_func",
    "fp8_token_wise_attn_func_with_fallback",
    "dynamically_quantize_fp8",
]

torch_sdpa = F.scaled_dot_product_attention
_NS = "quantumattention_amd"


def _define_composite_implicit_autograd_op(name, signature):
    def decorator(fn):
        torch.library.define(f"{_NS}::{name}", signature)
        torch.library.impl(f"{_NS}::{name}", ["CompositeImplicitAutograd"])(fn)
        return getattr(getattr(torch.ops, _NS), name)

    return decorator


def attn_func(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, *, scale=None) -> torch.Tensor:
    return attention(query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal, scale=scale)


@_define_composite_implicit_autograd_op(
    "attn_func_with_fallback",
    "(Tensor query, Tensor key, Tensor value, Tensor? attn_mask=None, float dropout_p=0.0, bool is_causal=False, *, float? scale=None) -> Tensor",
)
def attn_func_with_fallback(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, *, scale=None):
    supported, _ = can_use_attention(
        query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal, scale=scale)
    if supported:
        return attn_func(query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal, scale=scale)
    return torch_sdpa(query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal, scale=scale)


def fp8_attn_func(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, *, scale=None, scale_q=None,
                  scale_k=None, scaling_method: Optional[str] = None, amax_q=None, amax_k=None, ssq_q=None,
                  ssq_k=None) -> torch.Tensor:
    """quantum_attn_interface.py:101-127, plus amax_q / amax_k / ssq_q / ssq_k (keyword-only, optional): see nn.fp8_attention."""
    if scaling_method is None:
        scaling_method = "head-wise"
    return fp8_attention(query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal,
                         scale=scale, scale_q=scale_q, scale_k=scale_k, scaling_method=scaling_method, amax_q=amax_q, amax_k=amax_k,
                         ssq_q=ssq_q, ssq_k=ssq_k)


@_define_composite_implicit_autograd_op(
    "fp8_attn_func_with_fallback",
    "(Tensor query, Tensor key, Tensor value, Tensor? attn_mask=None, float dropout_p=0.0, bool is_causal=False, *, float? scale=None, str? scaling_method=None) -> Tensor",
)
def fp8_attn_func_with_fallback(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, *, scale=None,
                                scaling_method=None):
    if scaling_method is None:
        scaling_method = "head-wise"
    if can_use_attention(query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal,
                         scale=scale, scaling_method=scaling_method)[0]:
        return fp8_attn_func(query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal,
                             scale=scale, scaling_method=scaling_method)
    return torch_sdpa(query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal, scale=scale)


def fp8_token_wise_attn_func(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, *, scale=None,
                             scale_q=None, scale_k=None) -> torch.Tensor:
    return fp8_attention(query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal,
                         scale=scale, scale_q=scale_q, scale_k=scale_k, scaling_method="token-wise")


@_define_composite_implicit_autograd_op(
    "fp8_token_wise_attn_func_with_fallback",
    "(Tensor query, Tensor key, Tensor value, Tensor? attn_mask=None, float dropout_p=0.0, bool is_causal=False, *, float? scale=None) -> Tensor",
)
def fp8_token_wise_attn_func_with_fallback(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, *,
                                           scale=None):
    if can_use_attention(query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal,
                         scale=scale, scaling_method="token-wise")[0]:
        return fp8_token_wise_attn_func(query, key, value, attn_mask=attn_mask, dropout_p=dropout_p,
                                        is_causal=is_causal, scale=scale)
    return torch_sdpa(query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal, scale=scale)
