"""CPU oracle for the FP8 fused-attention hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this
package, and only as the checker.  The product (``quantumattention_amd``) never imports it and has no CPU
fallback for the HIP path.

Contents
--------
* ``qattn_oracle.c`` (bound here through ctypes): plain-C restatement of the reference's quantiser
  (``src/quantum_attn/nn.py:14-19``, both numerics) and of its op definition
  (``src/quantum_attn/ops.py:64-95`` and ``:17-29``) evaluated in fp64 on the same quantised inputs.
* ``torch_ref.py``: the reference's *literal* eager op (de-quantise in the output dtype, then aten SDPA)
  restated with torch CPU ops -- used as the timed ``cpu_baseline`` ("port") and pinned bit-for-bit to
  the golden ``o1_*`` vectors.

Pinned against ``tests/golden/*.npz`` (generated from the reference itself by
``tests/golden/gen_golden.py``) in ``tests/test_oracle_golden.py``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libqattn_oracle.so")

FMT_E4M3, FMT_E5M2, FMT_BF16, FMT_FP16 = 0, 1, 2, 3
FP8_MAX = {FMT_E4M3: 448.0, FMT_E5M2: 57344.0}


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (idempotent)."""
    src = os.path.join(_HERE, "qattn_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libqattn_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, i, l, f = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
        L.qo_quantize_fp8.argtypes = [vp, i, l, l, i, i, vp, vp]
        L.qo_quantize_fp8.restype = i
        L.qo_attention_forward.argtypes = [vp, vp, vp, i, i, i, vp, vp, vp, i, i, i, i, i, i, i, i, f, vp, vp]
        L.qo_attention_forward.restype = i
        L.qo_fp8_to_f32_array.argtypes = [vp, l, i, vp]
        L.qo_f32_to_fp8_array.argtypes = [vp, l, i, vp]
        L.qo_f32_to_bf16_array.argtypes = [vp, l, vp]
        L.qo_abi_version.restype = i
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


def fp8_to_f32(bytes_u8: np.ndarray, fmt: int = FMT_E4M3) -> np.ndarray:
    src = _c(bytes_u8, np.uint8)
    out = np.empty(src.shape, np.float32)
    lib().qo_fp8_to_f32_array(_ptr(src), src.size, fmt, _ptr(out))
    return out


def f32_to_fp8(x: np.ndarray, fmt: int = FMT_E4M3) -> np.ndarray:
    src = _c(x, np.float32)
    out = np.empty(src.shape, np.uint8)
    lib().qo_f32_to_fp8_array(_ptr(src), src.size, fmt, _ptr(out))
    return out


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    src = _c(x, np.float32)
    out = np.empty(src.shape, np.uint16)
    lib().qo_f32_to_bf16_array(_ptr(src), src.size, _ptr(out))
    return out


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(b, np.uint16).astype(np.uint32) << 16).view(np.float32)


def fp16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(b, np.uint16).view(np.float16).astype(np.float32)


def quantize_fp8(x_bits: np.ndarray, in_fmt: int, scale_mode: str = "head", fmt: int = FMT_E4M3,
                 numerics: str = "compiled"):
    """Restates nn.py:14-19.  ``x_bits``: uint16 bit patterns [B,H,S,D] of a bf16/fp16 tensor.

    Returns (payload uint8 [B,H,S,D], scale float32 [B,H] (head) or [B,H,S] (token))."""
    x = _c(x_bits, np.uint16)
    B, H, S, D = x.shape
    if scale_mode == "head":
        groups, inner, sshape = B * H, S * D, (B, H)
    elif scale_mode == "token":
        groups, inner, sshape = B * H * S, D, (B, H, S)
    else:
        raise ValueError(scale_mode)
    out = np.empty(x.shape, np.uint8)
    scale = np.empty(sshape, np.float32)
    rc = lib().qo_quantize_fp8(_ptr(x), in_fmt, groups, inner, fmt, {"compiled": 0, "eager": 1}[numerics],
                               _ptr(out), _ptr(scale))
    if rc != 0:
        raise RuntimeError(f"qo_quantize_fp8 failed: {rc}")
    return out, scale


def quantize_v_block(v_bits: np.ndarray, in_fmt: int, fmt: int = FMT_E4M3):
    """The build's block-scaled V (csrc/qattn_common.h vblock_exponent; the reference keeps V in 16 bit, tk/attention.py:286,318):
    every 64-key chunk of a head gets the power-of-two scale 2^e, e the smallest exponent with amax / 2^e <= fmax, found from the
    fp32 bits of the chunk's abs-max in integer arithmetic (amax = m 2^k, fmax = 1.75 2^Q: e = k - Q, one more if m > 1.75; zero,
    subnormal and non-finite abs-max: e = 0), and its elements are RNE-converted to fp8 after the exact division by 2^e.

    Returns (payload uint8 [B,H,S,D], E8M0 bytes uint8 [B,H,ceil(S/64)], the de-quantised tensor as bf16 bits [B,H,S,D] --
    fp8 x 2^e is exact in bf16, so attention_forward(..., v=those bits, v_fmt=FMT_BF16) attends exactly what the kernel does)."""
    x = _c(v_bits, np.uint16)
    B, H, S, D = x.shape
    f = (bf16_bits_to_f32(x) if in_fmt == FMT_BF16 else fp16_bits_to_f32(x)).astype(np.float32)
    nch = (S + 63) // 64
    pad = np.zeros((B, H, nch * 64, D), np.float32)
    pad[:, :, :S] = f
    ch = pad.reshape(B, H, nch, 64 * D)
    mag = np.abs(ch)
    amax = np.where(np.isnan(mag).any(-1), np.float32(np.nan), mag.max(-1)).astype(np.float32)   # a NaN in the chunk wins (packed integer max)
    bits = amax.view(np.uint32).astype(np.int64)
    ef = (bits >> 23) & 255
    e = ef - 127 - (8 if fmt == FMT_E4M3 else 15) + ((bits & 0x7FFFFF) > 0x600000)
    e = np.clip(np.where((ef == 0) | (ef == 255), 0, e), -126, 126).astype(np.int64)
    scale = np.ldexp(np.float32(1.0), e).astype(np.float32)[..., None]
    qmax = np.float32(448.0 if fmt == FMT_E4M3 else 57344.0)
    payload = f32_to_fp8(np.clip((ch / scale).astype(np.float32), -qmax, qmax), fmt).reshape(B, H, nch * 64, D)[:, :, :S]   # (clip keeps NaN)
    deq = (fp8_to_f32(payload, fmt).reshape(B, H, S, D) * np.repeat(scale, 64, axis=2).reshape(B, H, nch * 64, 1)[:, :, :S]).astype(np.float32)
    return payload, (e + 127).astype(np.uint8), f32_to_bf16_bits(deq)


def attention_forward(q, k, v, q_fmt, k_fmt, v_fmt, scale_q=None, scale_k=None, scale_v=None,
                      scale_mode: str = "head", causal: bool = False, sm_scale: float = 0.0,
                      return_lse: bool = False, q_offset: int = 0):
    """fp64-evaluated SDPA on the given (quantised) inputs; restates ops.py:64-95 / :17-29.
    q_offset (causal only): the query rows are rows q_offset .. of a longer causal problem (key j <= q_offset + i).

    q/k/v: numpy arrays of raw bits (uint8 for fp8 formats, uint16 for bf16/fp16), [B,H,S,D]."""
    dt = lambda fmt: np.uint8 if fmt in (FMT_E4M3, FMT_E5M2) else np.uint16
    q, k, v = _c(q, dt(q_fmt)), _c(k, dt(k_fmt)), _c(v, dt(v_fmt))
    B, Hq, Sq, D = q.shape
    _, Hkv, Skv, _ = k.shape
    assert v.shape == k.shape, (v.shape, k.shape)
    sq, sk, sv = _c(scale_q, np.float32), _c(scale_k, np.float32), _c(scale_v, np.float32)
    out = np.empty((B, Hq, Sq, D), np.float32)
    lse = np.empty((B, Hq, Sq), np.float32) if return_lse else None
    rc = lib().qo_attention_forward(_ptr(q), _ptr(k), _ptr(v), q_fmt, k_fmt, v_fmt, _ptr(sq), _ptr(sk), _ptr(sv),
                                    {"head": 0, "token": 1}[scale_mode], B, Hq, Hkv, Sq, Skv, D, (1 + int(q_offset)) if causal else 0,
                                    float(sm_scale), _ptr(out), _ptr(lse))
    if rc != 0:
        raise RuntimeError(f"qo_attention_forward failed: {rc}")
    return (out, lse) if return_lse else out
