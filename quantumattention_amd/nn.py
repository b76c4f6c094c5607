"""Dispatch + validation.  Mirrors src/quantum_attn/nn.py: `can_use_attention` (nn.py:282-307) returning
(bool, reason), `fp8_attention` (nn.py:433-539) raising ValueError(reason) on unsupported input, `attention`
(nn.py:325-391), `dynamically_quantize_fp8` (nn.py:22-42).  The reference wraps the call in torch.compile so that
Inductor can swap in its kernel; here the custom ops call the gfx950 kernels directly, so no compilation happens
on the hot path (calls from inside a user's torch.compile region trace the ops as opaque extern calls)."""
from typing import Optional, Tuple, Union

import torch
from torch import Tensor

from . import config
from .utils import checks

_HIP_SUPPORTED_HEAD_DIMS = [64, 128, 256]  # nn.py:45
_HIP_16BIT_HEAD_DIMS = [64, 128, 256]  # same set as the fp8 path (nn.py:45)
_FP8_DTYPES = (torch.float8_e4m3fn, torch.float8_e5m2)


def _ops():
    from . import ops  # registers the custom ops (imports the native binding lazily at call time)

    return ops


def _hip_supported_head_dim(n: Union[int, torch.SymInt]) -> bool:
    return n in _HIP_SUPPORTED_HEAD_DIMS


def _validate_hip_input(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None,
                        scaling_method=None) -> Tuple[bool, str]:
    """Same contract and messages as _validate_tk_tma_input (nn.py:52-124); differences: token-wise scaling and
    e5m2 are accepted (the reference routes token-wise to its Triton kernel, nn.py:135-205), GQA is accepted."""
    if any(t.requires_grad for t in (query, key, value)):
        return False, "NYI: query, key, and value must be leaf tensors"
    if attn_mask is not None:
        return False, "NYI: attn_mask must be None"
    if dropout_p != 0.0:
        return False, "NYI: dropout_p must be 0.0"
    if scale is not None:
        return False, "NYI: scale must be None"
    if scaling_method is None:
        # 16-bit sibling path (nn.py:325-391 -> ops.py:17-45): q, k, v share one 16-bit dtype; D in {64,128}
        if query.dtype not in (torch.float16, torch.bfloat16):
            return (
                False,
                f"Expected query to have dtype torch.float16 or torch.bfloat16, but got query.dtype: {query.dtype} instead.",
            )
        if query.dtype != value.dtype:
            return (
                False,
                f"Expected query and value to have the same dtype, but got query.dtype: {query.dtype}, value.dtype: {value.dtype} instead.",
            )
        if query.dim() == 4 and query.size(-1) not in _HIP_16BIT_HEAD_DIMS:
            return False, f"Unsupported head dimension: {query.size(-1)}"
    elif scaling_method not in ("head-wise", "token-wise"):
        return False, f"Unsupported scaling_method: {scaling_method}"
    if query.dtype not in (torch.float16, torch.bfloat16) + _FP8_DTYPES:
        return (
            False,
            f"Expected query to have dtype torch.float16, torch.bfloat16, or torch.float8_e4m3fn, but got query.dtype: {query.dtype} instead.",
        )
    if query.dtype != key.dtype:
        return (
            False,
            f"Expected query and key to have the same dtype, but got query.dtype: {query.dtype}, key.dtype: {key.dtype} instead.",
        )
    if value.dtype not in (torch.float16, torch.bfloat16):
        return (
            False,
            f"Expected value to have dtype torch.float16 or torch.bfloat16, but got value.dtype: {value.dtype} instead.",
        )
    if query.device != key.device or query.device != value.device:
        return (
            False,
            f"Expected query, key, and value to have the same device type, but got query.device: {query.device}, key.device: {key.device}, and value.device: {value.device} instead.",
        )
    if query.device.type != "cuda":
        return False, "Expected query, key, and value to be on a CUDA device"
    if query.dim() != 4 or key.dim() != 4 or value.dim() != 4:
        return False, "NYI: query, key, and value must be 4D tensors"
    if key.size(-2) != value.size(-2):
        return (
            False,
            f"Expect key and value to have the same sequence length but got Sk={key.size(-2)} and Sv={value.size(-2)}.",
        )
    if value.size(-1) != query.size(-1):
        return False, "NYI: query and value must have the same embedding dimension"
    if key.size(-3) != value.size(-3) or query.size(-3) % key.size(-3) != 0:
        return (
            False,
            f"Expect the number of query heads to be a multiple of the key/value heads but got Hq={query.size(-3)} and Hkv={key.size(-3)}.",
        )
    if not _hip_supported_head_dim(query.size(-1)):
        return False, f"Unsupported head dimension: {query.size(-1)}"
    return True, ""


@torch.compiler.assume_constant_result
def _pre_check_can_use_hip_attention(device):
    if device.type != "cuda":
        return False, f"Expected device to be on a CUDA device, but got device: {device} instead."
    if not config.attention.enable_hip_kernel:
        return False, "gfx950 HIP kernel is disabled"
    if not checks.is_gfx950(device):
        return False, "An AMD gfx950 (MI355X) device under PyTorch-ROCm is required"
    return True, ""


def can_use_hip_attention(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, *, scale=None,
                          scaling_method=None) -> Tuple[bool, str]:
    supported, reason = _pre_check_can_use_hip_attention(device=query.device)
    if not supported:
        return False, reason
    return _validate_hip_input(query, key, value, attn_mask, dropout_p, is_causal, scale, scaling_method=scaling_method)


def can_use_attention(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, *, scale=None,
                      scaling_method=None) -> Tuple[bool, str]:
    if checks.get_constant_attr("quantumattention_amd.config", "attention.skip_supported_check"):
        return True, ""
    supported, reason = can_use_hip_attention(
        query, key, value, attn_mask, dropout_p, is_causal, scale=scale, scaling_method=scaling_method)
    if supported:
        return True, ""
    return False, f"[hip_gfx950: {reason}]"


def _dynamically_quantize_fp8(t: torch.Tensor, *, reduction_dim=-1, fp8_dtype=torch.float8_e4m3fn):
    """Eager torch restatement used for CPU / fake tensors and shapes the HIP pre-pass does not cover
    (the reference's definition, nn.py:14-19)."""
    eps = torch.finfo(torch.float32).eps
    q_max = torch.finfo(fp8_dtype).max
    scale = t.abs().amax(reduction_dim, keepdim=True).mul(1.0 / q_max).clamp_min(eps)
    t_fp8 = (t / scale).clamp(-q_max, q_max).to(fp8_dtype)
    return t_fp8, scale.squeeze(reduction_dim).to(torch.float32)


def _norm_dims(reduction_dim, ndim):
    dims = reduction_dim if isinstance(reduction_dim, (list, tuple)) else [reduction_dim]
    return sorted(d % ndim for d in dims)


def dynamically_quantize_fp8(t: torch.Tensor, *, reduction_dim=-1) -> Tuple[torch.Tensor, torch.Tensor]:
    """nn.py:22-42.  4-D CUDA bf16/fp16 tensors reduced over the last dim (token-wise) or the last two
    (head-wise) run the HIP pre-pass (numerics selected by config.attention.quant_numerics, default = the
    reference's compiled-path numerics); anything else uses the eager torch definition."""
    from torch._subclasses.fake_tensor import is_fake

    dims = _norm_dims(reduction_dim, t.dim())
    hip_ok = (
        not is_fake(t) and t.is_cuda and t.dim() == 4 and t.dtype in (torch.float16, torch.bfloat16)
        and dims in ([3], [2, 3]) and t.size(-1) in _HIP_SUPPORTED_HEAD_DIMS and checks.is_gfx950(t.device)
        and config.attention.enable_hip_kernel
    )
    if not hip_ok:
        return _dynamically_quantize_fp8(t, reduction_dim=reduction_dim)
    return _ops().dynamically_quantize_fp8_op(t, dims == [3], config.attention.fp8_format,
                                              config.attention.quant_numerics)


def _attention_wrapper(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, *, scale=None) -> Tensor:
    """nn.py:310-322 -> quantum_attn::attention_forward (ops.py:17-45), here the bf16/fp16 MFMA kernel."""
    return _ops().attention_forward(query, key, value, attn_mask=attn_mask, dropout_p=dropout_p,
                                    is_causal=is_causal, scale=scale)


def attention(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, *, scale=None) -> Tensor:
    """nn.py:325-391."""
    supported, reason = can_use_attention(
        query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal, scale=scale)
    if not supported:
        raise ValueError(f"Unsupported input: {reason}")
    return _attention_wrapper(query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal,
                              scale=scale)


def _fp8_attention_wrapper(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, *, scale=None,
                           scale_q=None, scale_k=None, scaling_method=None) -> Tensor:
    """nn.py:394-430."""
    if (scale_q is None) != (scale_k is None):
        raise ValueError("scale_q and scale_k must be both provided or both not provided")
    ops = _ops()
    if scale_q is None:
        if scaling_method not in ("head-wise", "token-wise"):
            raise ValueError(f"Unsupported scaling_method: {scaling_method}")
        if query.dtype in _FP8_DTYPES:
            raise ValueError("fp8 query/key need scale_q and scale_k")
        return ops.fp8_quant_attention_forward(
            query, key, value, is_causal, scaling_method, config.attention.fp8_format,
            config.attention.quant_numerics, scale=scale)
    return ops.fp8_attention_forward(
        query, key, value, scale_q, scale_k, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal,
        scale=scale)


def fp8_attention(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, *, scale=None, scale_q=None,
                  scale_k=None, scaling_method=None) -> Tensor:
    """nn.py:433-539: validate (ValueError(reason) when unsupported), then run the wrapper."""
    supported, reason = can_use_attention(
        query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal, scale=scale,
        scaling_method=scaling_method)
    if not supported:
        raise ValueError(reason)
    if torch.compiler.is_dynamo_compiling():
        # mark head_dim and number of heads to be static (nn.py:484-488)
        for x in [query, key, value]:
            torch._dynamo.mark_static(x, -3)
            torch._dynamo.mark_static(x, -1)
    return _fp8_attention_wrapper(
        query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal, scale=scale,
        scale_q=scale_q, scale_k=scale_k, scaling_method=scaling_method)
