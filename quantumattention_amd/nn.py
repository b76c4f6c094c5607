"""Dispatch + validation.  Mirrors src/quantum_attn/nn.py: `can_use_attention` (nn.py:282-307) returning
(bool, reason), `fp8_attention` (nn.py:433-539) raising ValueError(reason) on unsupported input, `attention`
(nn.py:325-391), `dynamically_quantize_fp8` (nn.py:22-42).  The reference wraps the call in torch.compile so that
Inductor can swap in its kernel; here the custom ops call the gfx950 kernels directly, so no compilation happens
on the hot path (calls from inside a user's torch.compile region trace the ops as opaque extern calls).

`config.attention.force_eager_fallback` keeps the reference's meaning (nn.py:367-371, 503-516): skip the kernel and run
the op's eager *definition* (eager quantiser, de-quantise, aten SDPA -- ops.py:64-95) with torch ops on the tensors'
device.  It is a debugging switch the caller sets explicitly; nothing selects it silently."""
from typing import Callable, List, Optional, Tuple, Union

import torch
from torch import Tensor

from . import config
from .utils import checks

_HIP_SUPPORTED_HEAD_DIMS = [64, 128, 256]  # nn.py:45; the 16-bit sibling path covers the same set
_FP8_DTYPES = (torch.float8_e4m3fn, torch.float8_e5m2)
_HALF_DTYPES = (torch.float16, torch.bfloat16)


def _cfg(name: str):
    """config.attention.<name>, readable under torch.compile(fullgraph=True): dynamo does not trace the config proxy's
    __getattr__, so the lookup goes through an assume_constant_result helper as in the reference (utils/checks.py)."""
    return checks.config_value(f"attention.{name}")


def _ops():
    from . import ops  # registers the custom ops (imports the native binding lazily at call time)

    return ops


def _hip_supported_head_dim(n: Union[int, torch.SymInt]) -> bool:
    return n in _HIP_SUPPORTED_HEAD_DIMS


class _Call:
    """The arguments of one attention call, as the validation rules see them."""

    __slots__ = ("q", "k", "v", "attn_mask", "dropout_p", "scale", "scaling_method", "scale_q", "scale_k")

    def __init__(self, q, k, v, attn_mask, dropout_p, scale, scaling_method, scale_q=None, scale_k=None):
        self.q, self.k, self.v = q, k, v
        self.attn_mask, self.dropout_p, self.scale = attn_mask, dropout_p, scale
        self.scaling_method, self.scale_q, self.scale_k = scaling_method, scale_q, scale_k


# Each rule returns None (passes) or the reason string.  The reasons are interface contract: they are the strings
# `can_use_attention` hands back in the reference (nn.py:52-124) wherever the same condition exists there; rules the
# reference's C++ launcher enforces instead (batch / head_dim agreement, tk/attention.py:385-415) are added at the end.
def _r_leaf(c):
    if any(t.requires_grad for t in (c.q, c.k, c.v)):
        return "NYI: query, key, and value must be leaf tensors"


def _r_mask(c):
    if c.attn_mask is not None:
        return "NYI: attn_mask must be None"


def _r_dropout(c):
    if c.dropout_p != 0.0:
        return "NYI: dropout_p must be 0.0"


def _r_scale(c):
    if c.scale is not None:
        return "NYI: scale must be None"


def _r_method(c):
    if c.scaling_method is not None and c.scaling_method not in ("head-wise", "token-wise"):
        return f"Unsupported scaling_method: {c.scaling_method}"


def _r_query_dtype(c):
    if c.scaling_method is None:
        if c.q.dtype not in _HALF_DTYPES:
            return f"Expected query to have dtype torch.float16 or torch.bfloat16, but got query.dtype: {c.q.dtype} instead."
        if c.q.dtype != c.v.dtype:
            return (f"Expected query and value to have the same dtype, but got query.dtype: {c.q.dtype}, "
                    f"value.dtype: {c.v.dtype} instead.")
    elif c.q.dtype not in _HALF_DTYPES + _FP8_DTYPES:
        return ("Expected query to have dtype torch.float16, torch.bfloat16, or torch.float8_e4m3fn, "
                f"but got query.dtype: {c.q.dtype} instead.")


def _r_key_dtype(c):
    if c.q.dtype != c.k.dtype:
        return (f"Expected query and key to have the same dtype, but got query.dtype: {c.q.dtype}, "
                f"key.dtype: {c.k.dtype} instead.")


def _r_value_dtype(c):
    if c.v.dtype not in _HALF_DTYPES:
        return f"Expected value to have dtype torch.float16 or torch.bfloat16, but got value.dtype: {c.v.dtype} instead."


def _r_same_device(c):
    if c.q.device != c.k.device or c.q.device != c.v.device:
        return ("Expected query, key, and value to have the same device type, but got "
                f"query.device: {c.q.device}, key.device: {c.k.device}, and value.device: {c.v.device} instead.")


def _r_cuda(c):
    if c.q.device.type != "cuda":
        return "Expected query, key, and value to be on a CUDA device"


def _r_rank(c):
    if c.q.dim() != 4 or c.k.dim() != 4 or c.v.dim() != 4:
        return "NYI: query, key, and value must be 4D tensors"


def _r_kv_len(c):
    if c.k.size(-2) != c.v.size(-2):
        return f"Expect key and value to have the same sequence length but got Sk={c.k.size(-2)} and Sv={c.v.size(-2)}."


def _r_embed(c):
    if c.v.size(-1) != c.q.size(-1):
        return "NYI: query and value must have the same embedding dimension"


def _r_heads(c):  # GQA is accepted here (the reference requires Hq == Hkv at this level, nn.py:113-117)
    if c.k.size(-3) != c.v.size(-3) or c.q.size(-3) % c.k.size(-3) != 0:
        return ("Expect the number of query heads to be a multiple of the key/value heads but got "
                f"Hq={c.q.size(-3)} and Hkv={c.k.size(-3)}.")


def _r_head_dim(c):
    if not _hip_supported_head_dim(c.q.size(-1)):
        return f"Unsupported head dimension: {c.q.size(-1)}"


def _r_key_embed(c):  # tk/attention.py:394-396
    if c.k.size(-1) != c.q.size(-1):
        return (f"Expect query and key to have the same embedding dimension but got Dq={c.q.size(-1)} "
                f"and Dk={c.k.size(-1)}.")


def _r_batch(c):  # tk/attention.py:385-388
    if c.k.size(0) != c.q.size(0) or c.v.size(0) != c.q.size(0):
        return (f"Expect query, key, and value to have the same batch size but got Bq={c.q.size(0)}, "
                f"Bk={c.k.size(0)} and Bv={c.v.size(0)}.")


def _r_scales(c):  # tk/attention.py:402-414, extended to the token-wise shape [B,H,S]
    if (c.scale_q is None) != (c.scale_k is None):
        return "scale_q and scale_k must be both provided or both not provided"
    if c.scale_q is None:
        if c.q.dtype in _FP8_DTYPES:
            return "fp8 query and key need scale_q and scale_k"
        return None
    if c.q.dtype not in _FP8_DTYPES:
        return f"scale_q and scale_k are only accepted with fp8 query and key, but got query.dtype: {c.q.dtype}."
    return scale_shapes_reason(c.q, c.k, c.scale_q, c.scale_k)


_RULES: List[Callable[[_Call], Optional[str]]] = [
    _r_leaf, _r_mask, _r_dropout, _r_scale, _r_method, _r_query_dtype, _r_key_dtype, _r_value_dtype, _r_same_device,
    _r_cuda, _r_rank, _r_kv_len, _r_embed, _r_heads, _r_head_dim, _r_key_embed, _r_batch, _r_scales,
]


def scale_shapes_reason(query, key, scale_q, scale_k) -> Optional[str]:
    """fp32, same device, and exactly [B,H] (head-wise) or [B,H,S] (token-wise) for both scales -- else the reason."""
    for name, s, t in (("scale_q", scale_q, query), ("scale_k", scale_k, key)):
        if s.dtype != torch.float32:
            return f"Expected {name} to have dtype torch.float32, but got {s.dtype} instead."
        if s.device != t.device:
            return f"Expected {name} to be on {t.device}, but got {s.device} instead."
    head = (tuple(query.shape[:2]), tuple(key.shape[:2]))
    token = (tuple(query.shape[:3]), tuple(key.shape[:3]))
    got = (tuple(scale_q.shape), tuple(scale_k.shape))
    if got != head and got != token:
        return (f"Expected scale_q / scale_k of shape {head[0]} / {head[1]} (head-wise) or {token[0]} / {token[1]} "
                f"(token-wise), but got {got[0]} / {got[1]}.")
    return None


def _validate_hip_input(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None,
                        scaling_method=None, scale_q=None, scale_k=None) -> Tuple[bool, str]:
    """Contract of _validate_tk_tma_input (nn.py:52-124): first failing rule decides, in the reference's order.
    Differences: token-wise scaling and e5m2 are accepted (the reference routes token-wise to its Triton kernel,
    nn.py:135-205), GQA is accepted, and the launcher-level shape checks run here too so that a mismatching key /
    value / scale raises ValueError before any kernel launch."""
    call = _Call(query, key, value, attn_mask, dropout_p, scale, scaling_method, scale_q, scale_k)
    for rule in _RULES:
        reason = rule(call)
        if reason:
            return False, reason
    return True, ""


@torch.compiler.assume_constant_result
def _pre_check_can_use_hip_attention(device):
    if device.type != "cuda":
        return False, f"Expected device to be on a CUDA device, but got device: {device} instead."
    if not _cfg("enable_hip_kernel"):
        return False, "gfx950 HIP kernel is disabled"
    if not checks.is_gfx950(device):
        return False, "An AMD gfx950 (MI355X) device under PyTorch-ROCm is required"
    return True, ""


def can_use_hip_attention(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, *, scale=None,
                          scaling_method=None, scale_q=None, scale_k=None) -> Tuple[bool, str]:
    supported, reason = _pre_check_can_use_hip_attention(device=query.device)
    if not supported:
        return False, reason
    return _validate_hip_input(query, key, value, attn_mask, dropout_p, is_causal, scale, scaling_method=scaling_method,
                               scale_q=scale_q, scale_k=scale_k)


def can_use_attention(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, *, scale=None,
                      scaling_method=None, scale_q=None, scale_k=None) -> Tuple[bool, str]:
    if checks.config_value("attention.skip_supported_check"):
        return True, ""
    supported, reason = can_use_hip_attention(
        query, key, value, attn_mask, dropout_p, is_causal, scale=scale, scaling_method=scaling_method,
        scale_q=scale_q, scale_k=scale_k)
    if supported:
        return True, ""
    return False, f"[hip_gfx950: {reason}]"


def _fp8_dtype() -> torch.dtype:
    name = _cfg("fp8_format")
    if name not in ("e4m3", "e5m2"):
        raise ValueError(f"Unsupported config.attention.fp8_format: {name!r} (expected 'e4m3' or 'e5m2')")
    return torch.float8_e5m2 if name == "e5m2" else torch.float8_e4m3fn


def _dynamically_quantize_fp8(t: torch.Tensor, *, reduction_dim=-1, fp8_dtype=None):
    """Eager torch restatement used for CPU / fake tensors, shapes the HIP pre-pass does not cover and the
    force_eager_fallback switch (the reference's definition, nn.py:14-19)."""
    if fp8_dtype is None:
        fp8_dtype = _fp8_dtype()
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
    reference's compiled-path numerics); anything else uses the eager torch definition (same fp8 format)."""
    from torch._subclasses.fake_tensor import is_fake

    dims = _norm_dims(reduction_dim, t.dim())
    hip_ok = (
        not is_fake(t) and t.is_cuda and t.dim() == 4 and t.dtype in _HALF_DTYPES
        and dims in ([3], [2, 3]) and t.size(-1) in _HIP_SUPPORTED_HEAD_DIMS and checks.is_gfx950(t.device)
        and _cfg("enable_hip_kernel") and not _cfg("force_eager_fallback")
    )
    if not hip_ok:
        return _dynamically_quantize_fp8(t, reduction_dim=reduction_dim)
    return _ops().dynamically_quantize_fp8_op(t, dims == [3], _cfg("fp8_format"),
                                              _cfg("quant_numerics"))


def _expand_kv_heads(x: Tensor, hq: int) -> Tensor:
    return x if x.size(-3) == hq else x.repeat_interleave(hq // x.size(-3), dim=-3)


def _eager_attention(query, key, value, is_causal, scale) -> Tensor:
    """quantum_attn::attention_forward's eager definition (ops.py:17-29): aten SDPA."""
    hq = query.size(-3)
    return torch.nn.functional.scaled_dot_product_attention(
        query, _expand_kv_heads(key, hq), _expand_kv_heads(value, hq), is_causal=is_causal, scale=scale).contiguous()


def _eager_fp8_attention(query, key, value, scale_q, scale_k, is_causal, scale) -> Tensor:
    """quantum_attn::fp8_attention_forward's eager definition (ops.py:64-95): de-quantise q and k in value's dtype
    (scales broadcast over the trailing dims they were reduced over), then aten SDPA."""
    def dequant(x, s):
        s = s.to(value.dtype)
        while s.dim() < x.dim():
            s = s.unsqueeze(-1)
        return x.to(value.dtype) * s

    return _eager_attention(dequant(query, scale_q), dequant(key, scale_k), value, is_causal, scale)


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
    if _cfg("force_eager_fallback"):  # nn.py:367-371
        return _eager_attention(query, key, value, is_causal, scale)
    return _attention_wrapper(query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal,
                              scale=scale)


def _head_abs_max(t: Tensor) -> Tensor:
    """fp32 [B,H] max |x| over each head of a 16-bit [B,H,S,D] tensor (exact: the fp32 value of the largest 16-bit magnitude)."""
    return t.abs().amax(dim=(-2, -1)).to(torch.float32)


def _head_sum_sq(t: Tensor) -> Tensor:
    """fp32 [B,H] sum of x^2 over each head."""
    f = t.to(torch.float32)
    return (f * f).sum(dim=(-2, -1))


def _fp8_attention_wrapper(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, *, scale=None,
                           scale_q=None, scale_k=None, scaling_method=None, amax_q=None, amax_k=None, ssq_q=None,
                           ssq_k=None) -> Tensor:
    """nn.py:394-430."""
    if (scale_q is None) != (scale_k is None):
        raise ValueError("scale_q and scale_k must be both provided or both not provided")
    ops = _ops()
    if scale_q is None:
        if scaling_method not in ("head-wise", "token-wise"):
            raise ValueError(f"Unsupported scaling_method: {scaling_method}")
        if query.dtype in _FP8_DTYPES:
            raise ValueError("fp8 query/key need scale_q and scale_k")
        given = [t is not None for t in (amax_q, amax_k, ssq_q, ssq_k)]
        if any(given) and scaling_method != "head-wise":
            raise ValueError("amax_q / amax_k / ssq_q / ssq_k are per-head figures: head-wise scaling only")
        if (ssq_q is None) != (ssq_k is None):
            raise ValueError("ssq_q and ssq_k must be both provided or both not provided")
        amax_v = None
        if (torch.compiler.is_dynamo_compiling() and scaling_method == "head-wise" and not any(given)
                and _cfg("inline_abs_max_under_compile")):
            # The reference's quantiser is traced INTO the caller's graph (nn.py:410-418, 484-501) and Inductor fuses its abs-max
            # reduction with whatever produced query / key.  Same effect here: the per-head abs-max (and, for precision="auto", the
            # sums of squares of its score-spread estimate) are emitted as aten reductions in the caller's graph -- Inductor fuses
            # them into the producer's kernel -- and handed to the op, whose abs-max launch then has nothing left to read
            # (qattn_fp8_quant_attention_forward_ex).  abs().amax() of a 16-bit tensor is exact, so the scales and the output are
            # those of the eager call bit for bit (sums of squares: include/qattn.h on the dead-band edge).
            amax_q, amax_k = _head_abs_max(query), _head_abs_max(key)
            # V's abs-max only where V gets one scale per head: on the block-scaled paths (the headline one among them) the op never
            # reads it, and an op input cannot be eliminated as dead code -- the graph would carry a full extra read of V.
            # head_dim, dtype and the key length are static at trace time (mark_static above; a symbolic length counts as long).
            skv = key.shape[-2]
            if not isinstance(skv, int) or checks.fused_step_scales_v_per_head(query.shape[-1], query.dtype, scaling_method, skv):
                amax_v = _head_abs_max(value)
            if _cfg("precision") == "auto":
                ssq_q, ssq_k = _head_sum_sq(query), _head_sum_sq(key)
        return ops.fp8_quant_attention_forward(
            query, key, value, is_causal, scaling_method, _cfg("fp8_format"),
            _cfg("quant_numerics"), _cfg("precision"), amax_q, amax_k, ssq_q, ssq_k, amax_v, _cfg("output_layout"), scale=scale)
    if any(t is not None for t in (amax_q, amax_k, ssq_q, ssq_k)):
        raise ValueError("amax_q / amax_k describe 16-bit query / key; fp8 query / key come with scale_q / scale_k")
    return ops.fp8_attention_forward(
        query, key, value, scale_q, scale_k, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal,
        scale=scale)


def _fp8_attention_eager(query, key, value, is_causal, scale, scale_q, scale_k, scaling_method) -> Tensor:
    """force_eager_fallback: the reference's wrapper run eagerly (nn.py:503-516 -> :394-430 -> ops.py:64-95)."""
    if (scale_q is None) != (scale_k is None):
        raise ValueError("scale_q and scale_k must be both provided or both not provided")
    if scale_q is None:
        if scaling_method == "head-wise":        # nn.py:409-415: head-wise / token-wise / else raise, as the kernel path
            reduction_dim = [query.dim() - 2, query.dim() - 1]
        elif scaling_method == "token-wise":
            reduction_dim = query.dim() - 1
        else:
            raise ValueError(f"Unsupported scaling_method: {scaling_method}")
        query, scale_q = _dynamically_quantize_fp8(query, reduction_dim=reduction_dim)
        key, scale_k = _dynamically_quantize_fp8(key, reduction_dim=reduction_dim)
    return _eager_fp8_attention(query, key, value, scale_q, scale_k, is_causal, scale)


def fp8_attention(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, *, scale=None, scale_q=None,
                  scale_k=None, scaling_method=None, amax_q=None, amax_k=None, ssq_q=None, ssq_k=None) -> Tensor:
    """nn.py:433-539: validate (ValueError(reason) when unsupported), then run the wrapper.
    amax_q / amax_k (build extension, keyword-only): fp32 [B,H] per-head max |x| of 16-bit query / key from their producer --
    the quant pre-pass then skips its abs-max launch (the reference's Inductor fusion, nn.py:410-418).
    ssq_q / ssq_k (both or neither): fp32 [B,H] per-head sums of squares; with precision="auto" they stand in for the moments the
    skipped pass would have collected (without them heads with a wide score spread start one-term: same bound, other bits).
    Preconditions on amax_*: finite, >= the tensor's true abs-max (a smaller value clips; the sign is ignored; a NaN makes the head's
    scale NaN).  With only one of amax_q / amax_k and no ssq_* under precision="auto" nothing is saved (both tensors are still read for
    their sums of squares) and the result is the plain call's, bit for bit."""
    supported, reason = can_use_attention(
        query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal, scale=scale,
        scaling_method=scaling_method, scale_q=scale_q, scale_k=scale_k)
    if not supported:
        raise ValueError(reason)
    if torch.compiler.is_dynamo_compiling():
        # mark head_dim and number of heads to be static (nn.py:484-488)
        for x in [query, key, value]:
            torch._dynamo.mark_static(x, -3)
            torch._dynamo.mark_static(x, -1)
    elif _cfg("force_eager_fallback"):  # nn.py:503-516
        return _fp8_attention_eager(query, key, value, is_causal, scale, scale_q, scale_k, scaling_method)   # (recomputes the abs-max itself)
    return _fp8_attention_wrapper(
        query, key, value, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal, scale=scale,
        scale_q=scale_q, scale_k=scale_k, scaling_method=scaling_method, amax_q=amax_q, amax_k=amax_k, ssq_q=ssq_q, ssq_k=ssq_k)
