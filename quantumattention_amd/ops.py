"""Operator boundary.  Mirrors src/quantum_attn/ops.py: two torch custom ops with the reference's schemas
(ops.py:32-42, 98-110), registered for device type "cuda" (= HIP on PyTorch-ROCm) with fake impls so that
callers under torch.compile keep working.  The namespace is `quantumattention_amd` rather than `quantum_attn`
so the reference can be imported beside this package in one process (torch.library.define would collide).

Unlike the reference -- whose eager impl is aten SDPA on de-quantised inputs and whose real kernel is reached
only through an Inductor lowering (inductor/kernels/attention.py:1037-1065) -- the implementation here calls
the hand-written gfx950 kernels through the C ABI directly.  There is no eager/CPU fallback at this level.
"""
from typing import Optional

import torch

from . import _native, config

_custom_op = torch.library.custom_op
_register_fake = torch.library.register_fake


def _out_like(query: torch.Tensor, value: torch.Tensor, output_layout: str = "contiguous") -> torch.Tensor:
    return _native.empty_output(query, value.dtype, output_layout)


def _check_op_args(attn_mask, dropout_p, scale):
    # same asserts as tk_fp8_attention_forward_kernel (inductor/kernels/attention.py:55-73)
    if attn_mask is not None:
        raise RuntimeError("attn_mask is not supported by the gfx950 attention kernel")
    if dropout_p != 0.0:
        raise RuntimeError("dropout_p must be 0.0 for the gfx950 attention kernel")


@_custom_op("quantumattention_amd::attention_forward", mutates_args=(), device_types=("cuda",))
def attention_forward(
    query: torch.Tensor,
    key: torch.Tensor,
    value: torch.Tensor,
    attn_mask: Optional[torch.Tensor] = None,
    dropout_p: float = 0.0,
    is_causal: bool = False,
    *,
    scale: Optional[float] = None,
) -> torch.Tensor:
    """16-bit sibling op, same schema as quantum_attn::attention_forward (ops.py:32-42): query/key/value bf16 or fp16
    [B,H,S,D]; K and V are re-laid into MFMA fragment order, then the bf16/fp16 MFMA kernel runs."""
    _check_op_args(attn_mask, dropout_p, scale)
    if query.dtype not in (torch.bfloat16, torch.float16) or key.dtype != query.dtype or value.dtype != query.dtype:
        raise RuntimeError(f"query/key/value must share bf16 or fp16, got {query.dtype}, {key.dtype}, {value.dtype}")
    _, _, Hkv, _, Skv, _ = _native._check_qkv(query, key, value)
    k_frag = _native.pack16(key, _native.LAYOUT_K16FRAG)
    v_frag = _native.pack16(value, _native.LAYOUT_V16FRAG)
    return _native.attention_forward_16(query, k_frag, v_frag, Hkv=Hkv, Skv=Skv, is_causal=is_causal,
                                        sm_scale=0.0 if scale is None else float(scale),
                                        fast_exp=bool(config.attention.fast_exp16), output_layout=config.attention.output_layout)


@_register_fake("quantumattention_amd::attention_forward")
def _(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, *, scale=None):
    return _out_like(query, value, config.attention.output_layout)


@_custom_op("quantumattention_amd::fp8_attention_forward", mutates_args=(), device_types=("cuda",))
def fp8_attention_forward(
    query: torch.Tensor,
    key: torch.Tensor,
    value: torch.Tensor,
    scale_q: Optional[torch.Tensor] = None,
    scale_k: Optional[torch.Tensor] = None,
    attn_mask: Optional[torch.Tensor] = None,
    dropout_p: float = 0.0,
    is_causal: bool = False,
    *,
    scale: Optional[float] = None,
) -> torch.Tensor:
    """query/key: fp8 (e4m3fn or e5m2) [B,H,S,D] row-major with fp32 scales [B,H] (head-wise) or [B,H,S]
    (token-wise); value: bf16/fp16.  Same contract as quantum_attn::fp8_attention_forward (ops.py:98-121)."""
    _check_op_args(attn_mask, dropout_p, scale)
    if scale_q is None or scale_k is None:
        raise RuntimeError("fp8_attention_forward needs scale_q and scale_k")
    if query.dtype not in (torch.float8_e4m3fn, torch.float8_e5m2) or key.dtype != query.dtype:
        raise RuntimeError(f"query/key must share an fp8 dtype, got {query.dtype} and {key.dtype}")
    if value.dtype not in (torch.bfloat16, torch.float16):
        raise RuntimeError(f"value must be bf16 or fp16, got {value.dtype}")
    if query.dim() != 4 or key.dim() != 4 or value.dim() != 4:
        raise RuntimeError("query, key and value must be 4-D")
    B, Hq, Sq, D = query.shape
    if key.shape != value.shape or key.shape[0] != B or key.shape[3] != D or key.device != query.device or value.device != query.device:
        raise RuntimeError(f"key {tuple(key.shape)} / value {tuple(value.shape)} do not match query {tuple(query.shape)} "
                           "(same batch, head_dim, device; key and value the same shape)")
    Hkv, Skv = key.shape[1], key.shape[2]
    if Hkv == 0 or Hq % Hkv != 0:
        raise RuntimeError(f"Hq={Hq} is not a multiple of Hkv={Hkv}")
    # scale_q / scale_k: fp32 on the same device, exactly [B,H] (head-wise) or [B,H,S] (token-wise); tk/attention.py:402-414
    from .nn import scale_shapes_reason

    reason = scale_shapes_reason(query, key, scale_q, scale_k)
    if reason:
        raise RuntimeError(reason)
    if config.attention.pv_precision not in ("fp8", "16bit"):
        raise ValueError(f"Unsupported config.attention.pv_precision: {config.attention.pv_precision!r} (expected 'fp8' or '16bit')")
    # ONE C call with the pybind function's contract (tk/attention.py:357-360): the K re-lay and the V handling happen inside, in a
    # workspace.  pv_precision = "16bit": the reference kernel's own numerics, value stays 16-bit and P is cast to 16 bit
    # (tk/attention.py:72,286,318); "fp8" (default): V quantised head-wise, both GEMMs on FP8 MFMA.
    return _native.fp8_attention_forward_rowmajor(
        query, key, value, scale_q, scale_k, is_causal=is_causal, pv_16bit=config.attention.pv_precision == "16bit",
        sm_scale=0.0 if scale is None else float(scale), precision=config.attention.precision)


@_register_fake("quantumattention_amd::fp8_attention_forward")
def _(query, key, value, scale_q=None, scale_k=None, attn_mask=None, dropout_p=0.0, is_causal=False, *, scale=None):
    return _out_like(query, value)


@_custom_op("quantumattention_amd::fp8_quant_attention_forward", mutates_args=(), device_types=("cuda",))
def fp8_quant_attention_forward(
    query: torch.Tensor,
    key: torch.Tensor,
    value: torch.Tensor,
    is_causal: bool = False,
    scaling_method: str = "head-wise",
    fp8_format: str = "e4m3",
    numerics: str = "compiled",
    precision: str = "auto",
    amax_q: Optional[torch.Tensor] = None,
    amax_k: Optional[torch.Tensor] = None,
    ssq_q: Optional[torch.Tensor] = None,
    ssq_k: Optional[torch.Tensor] = None,
    amax_v: Optional[torch.Tensor] = None,
    output_layout: str = "contiguous",
    *,
    scale: Optional[float] = None,
) -> torch.Tensor:
    """Fused entry for 16-bit inputs: the quant pre-pass (nn.py:410-418) writes K and V straight into the MFMA
    fragment layouts, then the attention kernel runs -- what `_fp8_attention_wrapper` (nn.py:394-430) does in the
    reference through Inductor, without the intermediate row-major K copy.  amax_q / amax_k (fp32 [B,H], head-wise only):
    per-head max |x| of query / key from the kernel that produced them; the abs-max launch then has nothing to read --
    the hand-off the reference gets from Inductor fusing the quantiser into the producer.  ssq_q / ssq_k (fp32 [B,H], both or
    neither): per-head sums of squares, which keep precision="auto" its score-spread estimate when the abs-max pass is skipped.
    amax_v (fp32 [B,Hkv]): the same for value, read only where V has one scale per head (token-wise scales, or more than 16384 keys per
    head; elsewhere the fused step scales V per 64-key chunk inside the quantise pass and needs no abs-max of it: include/qattn.h PATH TABLE)."""
    return _native.fp8_quant_attention_forward(
        query, key, value, is_causal=is_causal, scaling=scaling_method, fp8_dtype=_native.fp8_dtype_of(fp8_format),
        numerics=numerics, sm_scale=0.0 if scale is None else float(scale), precision=precision, amax_q=amax_q, amax_k=amax_k,
        amax_v=amax_v, ssq_q=ssq_q, ssq_k=ssq_k, output_layout=output_layout)


@_register_fake("quantumattention_amd::fp8_quant_attention_forward")
def _(query, key, value, is_causal=False, scaling_method="head-wise", fp8_format="e4m3", numerics="compiled",
      precision="auto", amax_q=None, amax_k=None, ssq_q=None, ssq_k=None, amax_v=None, output_layout="contiguous", *, scale=None):
    return _out_like(query, value, output_layout)


@_custom_op("quantumattention_amd::dynamically_quantize_fp8", mutates_args=(), device_types=("cuda",))
def dynamically_quantize_fp8_op(t: torch.Tensor, token_wise: bool, fp8_format: str = "e4m3",
                                numerics: str = "compiled") -> tuple[torch.Tensor, torch.Tensor]:
    """HIP quant pre-pass for a 4-D [B,H,S,D] tensor (nn.py:14-19): head-wise (dims 2,3) or token-wise (dim 3)."""
    return _native.quant_fp8(t, scaling="token-wise" if token_wise else "head-wise",
                             fp8_dtype=_native.fp8_dtype_of(fp8_format), layout=_native.LAYOUT_ROWMAJOR,
                             numerics=numerics)


@_register_fake("quantumattention_amd::dynamically_quantize_fp8")
def _(t, token_wise, fp8_format="e4m3", numerics="compiled"):
    q = torch.empty(t.shape, dtype=_native.fp8_dtype_of(fp8_format), device=t.device)
    s = torch.empty(t.shape[:3] if token_wise else t.shape[:2], dtype=torch.float32, device=t.device)
    return q, s
