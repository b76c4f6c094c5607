"""ctypes binding of libqattn_hip.so (include/qattn.h).  No CPU fallback: a missing library is a hard error."""
import ctypes
import math
import os
from typing import Optional, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (QUANTUM_ATTN_LIBRARY: another build of the same library, e.g. a tuning variant from `build.py --variant=...`; development aid)
LIB_PATH = os.environ.get("QUANTUM_ATTN_LIBRARY") or os.path.join(_HERE, "libqattn_hip.so")

FMT_E4M3, FMT_E5M2, FMT_BF16, FMT_FP16 = 0, 1, 2, 3
SCALE_HEAD, SCALE_TOKEN = 0, 1
LAYOUT_ROWMAJOR, LAYOUT_KFRAG, LAYOUT_VFRAG, LAYOUT_K16FRAG, LAYOUT_V16FRAG = 0, 1, 2, 3, 4
NUMERICS = {"compiled": 0, "eager": 1}
ABI_VERSION = 8
PRECISION = {"auto": 0, "fast": 1, "accurate": 2}
LSE_NATURAL, LSE_REFERENCE = 0, 1
PATH_ONE_TERM, PATH_TWO_TERM, PATH_V16 = 0, 1, 2   # QATTN_PATH_*: the fused entry's per-row debug output

_FMT_OF_DTYPE = {
    torch.float8_e4m3fn: FMT_E4M3,
    torch.float8_e5m2: FMT_E5M2,
    torch.bfloat16: FMT_BF16,
    torch.float16: FMT_FP16,
}
FP8_DTYPE = {"e4m3": torch.float8_e4m3fn, "e5m2": torch.float8_e5m2}

EXPORTS = (
    "qattn_abi_version", "qattn_strerror", "qattn_check_device", "qattn_fp8_tensor_bytes",
    "qattn_quant_workspace_bytes", "qattn_quant_fp8", "qattn_quant_qkv_workspace_bytes", "qattn_quant_qkv_fp8",
    "qattn_pack_fp8", "qattn_fp8_attention_forward",
    "qattn_16bit_tensor_bytes", "qattn_pack16", "qattn_attention_forward_16", "qattn_fp8_quant_attention_forward",
    "qattn_attention_workspace_bytes", "qattn_lse_row_stride", "qattn_fp8_quant_attention_workspace_bytes",
    "qattn_profile_attention", "qattn_last_attention_ms", "qattn_vblock_exponent", "qattn_fp8_quant_attention_forward_ex",
    "qattn_attention_stamp_bytes", "qattn_fp8_quant_attention_forward_stamped", "qattn_mfma_probe_bytes", "qattn_mfma_probe",
    "qattn_fp8_attention_rowmajor_workspace_bytes", "qattn_fp8_attention_forward_rowmajor", "qattn_describe_path",
    "qattn_fp8_quant_attention_forward_strided", "qattn_pack16_strided", "qattn_attention_forward_16_strided",
)


class PathDesc(ctypes.Structure):
    """qattn_path_desc (include/qattn.h): what an entry runs for given arguments."""
    _fields_ = [(n, ctypes.c_int) for n in ("kernel", "q_quant", "v_format", "sweep_p", "precise", "early", "start_mode", "lse")]


ENTRY = {"separate": 0, "separate16": 1, "fused": 2}
# names of the enum values of qattn_path_desc's fields, in the order of their codes (the vocabulary of the header's path table)
PATH_DESC_NAMES = {
    "kernel": ("v2", "v4", "pv16"), "q_quant": ("caller", "prepass", "kernel"), "v_format": ("head", "block", "16bit"),
    "sweep_p": ("byte", "exact", "p16"), "precise": ("two-term", "v16", "none"), "early": ("two-term", "v16-inline", "v16-launch", "none"),
    "start_mode": ("keys", "moments", "none"), "lse": ("exact", "quantised"),
}

_lib = None


def lib() -> ctypes.CDLL:
    """Load the library (once).  Raises if it has not been built -- the HIP path has no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m quantumattention_amd.build` "
            "(hipcc --offload-arch=gfx950). The gfx950 FP8 attention path has no CPU or eager fallback."
        )
    if os.path.abspath(LIB_PATH) != os.path.join(_HERE, "libqattn_hip.so"):
        # never silent: a stale variant with the right ABI number would otherwise be timed or shipped unnoticed
        import warnings

        warnings.warn(f"quantumattention_amd: loading a NON-DEFAULT kernel library {LIB_PATH} (QUANTUM_ATTN_LIBRARY / LIB_PATH "
                      "override; development aid)", RuntimeWarning, stacklevel=2)
    L = ctypes.CDLL(LIB_PATH)
    vp, i, f, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
    L.qattn_abi_version.restype = i
    L.qattn_strerror.restype = ctypes.c_char_p
    L.qattn_strerror.argtypes = [i]
    L.qattn_check_device.restype = i
    L.qattn_vblock_exponent.restype = i
    L.qattn_vblock_exponent.argtypes = [ctypes.c_uint, i]
    L.qattn_fp8_tensor_bytes.restype = sz
    L.qattn_fp8_tensor_bytes.argtypes = [i, i, i, i, i]
    L.qattn_quant_workspace_bytes.restype = sz
    L.qattn_quant_workspace_bytes.argtypes = [i, i, i, i, i]
    L.qattn_quant_fp8.restype = i
    L.qattn_quant_fp8.argtypes = [vp, i, vp, vp, i, i, i, i, i, i, i, i, vp, sz, vp]
    L.qattn_quant_qkv_workspace_bytes.restype = sz
    L.qattn_quant_qkv_workspace_bytes.argtypes = [i, i, i]
    L.qattn_quant_qkv_fp8.restype = i
    L.qattn_quant_qkv_fp8.argtypes = [vp, vp, vp, i, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, vp, sz, vp]
    L.qattn_pack_fp8.restype = i
    L.qattn_pack_fp8.argtypes = [vp, vp, i, i, i, i, i, vp]
    L.qattn_attention_workspace_bytes.restype = sz
    L.qattn_attention_workspace_bytes.argtypes = [i, i, i]
    L.qattn_lse_row_stride.restype = sz
    L.qattn_lse_row_stride.argtypes = [i, i]
    L.qattn_fp8_attention_forward.restype = i
    L.qattn_fp8_attention_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, i, i, f, i, i, vp, sz, vp]
    L.qattn_16bit_tensor_bytes.restype = sz
    L.qattn_16bit_tensor_bytes.argtypes = [i, i, i, i, i]
    L.qattn_pack16.restype = i
    L.qattn_pack16.argtypes = [vp, vp, i, i, i, i, i, vp]
    L.qattn_attention_forward_16.restype = i
    L.qattn_attention_forward_16.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, f, i, vp]
    L.qattn_profile_attention.restype = None
    L.qattn_profile_attention.argtypes = [i]
    L.qattn_last_attention_ms.restype = ctypes.c_float
    L.qattn_fp8_quant_attention_workspace_bytes.restype = sz
    L.qattn_fp8_quant_attention_workspace_bytes.argtypes = [i, i, i, i]
    L.qattn_fp8_quant_attention_forward.restype = i
    L.qattn_fp8_quant_attention_forward.argtypes = [vp, vp, vp, i, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, i, f, i, vp, sz, vp]
    L.qattn_fp8_quant_attention_forward_ex.restype = i
    L.qattn_fp8_quant_attention_forward_ex.argtypes = [vp, vp, vp, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                                       i, i, i, i, i, i, i, i, i, i, f, i, vp, i, vp, vp, sz, vp]
    L.qattn_pack16_strided.restype = i
    L.qattn_pack16_strided.argtypes = [vp, vp, vp, i, i, i, i, i, vp]
    L.qattn_attention_forward_16_strided.restype = i
    L.qattn_attention_forward_16_strided.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, f, i, vp]
    L.qattn_fp8_quant_attention_forward_strided.restype = i
    L.qattn_fp8_quant_attention_forward_strided.argtypes = [vp, vp, vp, vp, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                                            i, i, i, i, i, i, i, i, i, i, f, i, vp, i, vp, vp, sz, vp]
    L.qattn_attention_stamp_bytes.restype = sz
    L.qattn_attention_stamp_bytes.argtypes = [i, i, i]
    L.qattn_fp8_quant_attention_forward_stamped.restype = i
    L.qattn_fp8_quant_attention_forward_stamped.argtypes = [vp, vp, vp, i, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, i, f, i, vp, sz, vp, sz, vp]
    L.qattn_mfma_probe_bytes.restype = sz
    L.qattn_mfma_probe_bytes.argtypes = []
    L.qattn_mfma_probe.restype = i
    L.qattn_mfma_probe.argtypes = [vp, sz, i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int), vp]
    L.qattn_fp8_attention_rowmajor_workspace_bytes.restype = sz
    L.qattn_fp8_attention_rowmajor_workspace_bytes.argtypes = [i, i, i, i, i, i]
    L.qattn_fp8_attention_forward_rowmajor.restype = i
    L.qattn_fp8_attention_forward_rowmajor.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, i, i, f, i, i, vp, sz, vp]
    L.qattn_describe_path.restype = i
    L.qattn_describe_path.argtypes = [i, i, i, i, i, i, ctypes.POINTER(PathDesc)]
    if L.qattn_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libqattn_hip.so ABI {L.qattn_abi_version()} != expected {ABI_VERSION}; rebuild it")
    _lib = L
    return L


def _require(cond: bool, what: str) -> None:
    """Argument checks of the binding raise (they must survive `python -O`, unlike assert)."""
    if not cond:
        raise ValueError(what)


def _scale_mode(scaling: str) -> int:
    _require(scaling in ("head-wise", "token-wise"), f"Unsupported scaling_method: {scaling}")
    return SCALE_HEAD if scaling == "head-wise" else SCALE_TOKEN


def _precision(name) -> int:
    _require(name in PRECISION, f"Unsupported precision: {name!r} (expected one of {sorted(PRECISION)})")
    return PRECISION[name]


def _numerics(name) -> int:
    _require(name in NUMERICS, f"Unsupported quant_numerics: {name!r} (expected one of {sorted(NUMERICS)})")
    return NUMERICS[name]


def fp8_dtype_of(name) -> torch.dtype:
    """config.attention.fp8_format / the ops' fp8_format argument -> torch dtype; anything else is a ValueError, not a KeyError."""
    _require(name in FP8_DTYPE, f"Unsupported fp8_format: {name!r} (expected one of {sorted(FP8_DTYPE)})")
    return FP8_DTYPE[name]


def _check_qkv(q, k, v):
    """q [B,Hq,Sq,D], k / v [B,Hkv,Skv,D] on one device with one 16-bit dtype (tk/attention.py:362-400)."""
    _require(q.is_cuda and q.dim() == 4 and k.dim() == 4 and v.dim() == 4, "query, key and value must be 4-D device tensors")
    _require(q.device == k.device == v.device, "query, key and value must be on the same device")
    _require(q.dtype == k.dtype == v.dtype, f"query, key and value must share a dtype, got {q.dtype}, {k.dtype}, {v.dtype}")
    B, Hq, Sq, D = q.shape
    _require(k.shape == v.shape, f"key and value shapes differ: {tuple(k.shape)} vs {tuple(v.shape)}")
    _require(k.shape[0] == B and k.shape[3] == D, f"key/value batch or head_dim {tuple(k.shape)} do not match the query {tuple(q.shape)}")
    Hkv, Skv = k.shape[1], k.shape[2]
    _require(Hkv > 0 and Hq % Hkv == 0, f"Hq={Hq} is not a multiple of Hkv={Hkv}")
    return B, Hq, Hkv, Sq, Skv, D


def _strided_ok(t: torch.Tensor) -> bool:
    """A 4-D view the kernels address directly (include/qattn_strided.h): head_dim innermost and dense, every other stride a non-negative
    multiple of 8 elements (rows 16-byte aligned), rows at most 2^23 elements apart, base 16-byte aligned."""
    if t.is_contiguous():
        return True
    D = t.shape[3]
    return (t.stride(3) == 1 and all(t.shape[i] == 1 or (t.stride(i) >= 0 and t.stride(i) % 8 == 0) for i in (0, 1, 2))
            and (t.shape[2] == 1 or D <= t.stride(2) <= 2 ** 23) and t.data_ptr() % 16 == 0)


def empty_output(query: torch.Tensor, dtype, output_layout: str = "contiguous") -> torch.Tensor:
    """The output tensor of an attention call on `query` [B,H,S,D]: dense (the reference: tk/attention.py:434-437), or -- output_layout
    "like_query" and a query whose (b, s, h, d) order is dense in memory, i.e. x.view(B, S, H, D).transpose(1, 2) -- the transposed view
    of a dense [B,S,H,D] tensor (config.attention.output_layout)."""
    _require(output_layout in ("contiguous", "like_query"), f"output_layout must be 'contiguous' or 'like_query', got {output_layout!r}")
    B, H, S, D = query.shape
    if output_layout == "like_query" and not query.is_contiguous() and query.transpose(1, 2).is_contiguous():
        return torch.empty((B, S, H, D), dtype=dtype, device=query.device).transpose(1, 2)
    return torch.empty((B, H, S, D), dtype=dtype, device=query.device)


def _strides3(t: torch.Tensor):
    """{batch, head, row} element strides of a view `_strided_ok` accepted, as the C entries take them (None: dense)."""
    if t.is_contiguous():
        return None
    H, S, D = t.shape[1], t.shape[2], t.shape[3]
    dense = (H * S * D, S * D, D)     # (a dimension of size 1 has no stride of its own)
    return (ctypes.c_longlong * 3)(*[t.stride(i) if t.shape[i] > 1 else dense[i] for i in (0, 1, 2)])


def _check_scales(scale_q, scale_k, scale_v, mode, B, Hq, Hkv, Sq, Skv, device) -> None:
    """fp32, on `device`, exactly [B,H] (head-wise) or [B,H,S] (token-wise): tk/attention.py:402-414."""
    want_q = (B, Hq) if mode == SCALE_HEAD else (B, Hq, Sq)
    want_k = (B, Hkv) if mode == SCALE_HEAD else (B, Hkv, Skv)
    for name, s, want in (("scale_q", scale_q, want_q), ("scale_k", scale_k, want_k), ("scale_v", scale_v, (B, Hkv))):
        if s is None and name == "scale_v":
            continue
        _require(s is not None and s.dtype == torch.float32 and s.device == device and tuple(s.shape) == want,
                 f"{name} must be float32 {want} on {device}")


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed: {lib().qattn_strerror(rc).decode()} (code {rc})")


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def fmt_of(dtype: torch.dtype) -> int:
    try:
        return _FMT_OF_DTYPE[dtype]
    except KeyError:
        raise ValueError(f"unsupported dtype {dtype}") from None


def quant_fp8(x: torch.Tensor, *, scaling: str = "head-wise", fp8_dtype=torch.float8_e4m3fn,
              layout: int = LAYOUT_ROWMAJOR, numerics: str = "compiled") -> Tuple[torch.Tensor, torch.Tensor]:
    """bf16/fp16 [B,H,S,D] -> (fp8 payload, fp32 scale).  Row-major payloads come back as an fp8 tensor of
    x's shape; fragment layouts as a flat uint8 buffer (their layout is private to the library)."""
    _require(x.is_cuda and x.dim() == 4, "quant_fp8 needs a 4-D device tensor")
    x = x.contiguous()
    B, H, S, D = x.shape
    L = lib()
    mode = _scale_mode(scaling)
    nbytes = L.qattn_fp8_tensor_bytes(layout, B, H, S, D)
    with torch.cuda.device(x.device):
        if layout == LAYOUT_ROWMAJOR:
            out = torch.empty((B, H, S, D), dtype=fp8_dtype, device=x.device)
        else:
            out = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
        scale = torch.empty((B, H) if mode == SCALE_HEAD else (B, H, S), dtype=torch.float32, device=x.device)
        ws_bytes = L.qattn_quant_workspace_bytes(B, H, S, D, mode)
        ws = torch.empty((max(ws_bytes, 4),), dtype=torch.uint8, device=x.device)
        rc = L.qattn_quant_fp8(x.data_ptr(), fmt_of(x.dtype), out.data_ptr(), scale.data_ptr(), B, H, S, D,
                               fmt_of(fp8_dtype), mode, _numerics(numerics), layout, ws.data_ptr(), ws_bytes,
                               _stream(x))
    _check(rc, "qattn_quant_fp8")
    return out, scale


def quant_qkv_fp8(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, scaling: str = "head-wise",
                  fp8_dtype=torch.float8_e4m3fn, numerics: str = "compiled"):
    """Fused pre-pass: (q8 row-major fp8, k_frag, v_frag, scale_q, scale_k, scale_v) in two launches."""
    B, Hq, Hkv, Sq, Skv, D = _check_qkv(q, k, v)
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    L = lib()
    mode = _scale_mode(scaling)
    dev = q.device
    with torch.cuda.device(dev):
        q8 = torch.empty((B, Hq, Sq, D), dtype=fp8_dtype, device=dev)
        kf = torch.empty((L.qattn_fp8_tensor_bytes(LAYOUT_KFRAG, B, Hkv, Skv, D),), dtype=torch.uint8, device=dev)
        vf = torch.empty((L.qattn_fp8_tensor_bytes(LAYOUT_VFRAG, B, Hkv, Skv, D),), dtype=torch.uint8, device=dev)
        sq = torch.empty((B, Hq) if mode == SCALE_HEAD else (B, Hq, Sq), dtype=torch.float32, device=dev)
        sk = torch.empty((B, Hkv) if mode == SCALE_HEAD else (B, Hkv, Skv), dtype=torch.float32, device=dev)
        sv = torch.empty((B, Hkv), dtype=torch.float32, device=dev)
        ws_bytes = L.qattn_quant_qkv_workspace_bytes(B, Hq, Hkv)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        rc = L.qattn_quant_qkv_fp8(q.data_ptr(), k.data_ptr(), v.data_ptr(), fmt_of(q.dtype), q8.data_ptr(),
                                   kf.data_ptr(), vf.data_ptr(), sq.data_ptr(), sk.data_ptr(), sv.data_ptr(),
                                   B, Hq, Hkv, Sq, Skv, D, fmt_of(fp8_dtype), mode, _numerics(numerics),
                                   ws.data_ptr(), ws_bytes, _stream(q))
    _check(rc, "qattn_quant_qkv_fp8")
    return q8, kf, vf, sq, sk, sv


def pack_fp8(x8: torch.Tensor, layout: int) -> torch.Tensor:
    """row-major fp8 [B,H,S,D] -> flat uint8 buffer in KFRAG / VFRAG layout."""
    _require(x8.is_cuda and x8.dim() == 4 and x8.dtype.itemsize == 1, "pack_fp8 needs a 4-D fp8 device tensor")
    x8 = x8.contiguous()
    B, H, S, D = x8.shape
    L = lib()
    with torch.cuda.device(x8.device):
        out = torch.empty((L.qattn_fp8_tensor_bytes(layout, B, H, S, D),), dtype=torch.uint8, device=x8.device)
        rc = L.qattn_pack_fp8(x8.data_ptr(), out.data_ptr(), B, H, S, D, layout, _stream(x8))
    _check(rc, "qattn_pack_fp8")
    return out


def fp8_attention_forward(q8: torch.Tensor, k_frag: torch.Tensor, v_frag: torch.Tensor, scale_q: torch.Tensor,
                          scale_k: torch.Tensor, scale_v: Optional[torch.Tensor], *, Hkv: int, Skv: int,
                          out_dtype: torch.dtype, is_causal: bool, scaling: str = "head-wise",
                          sm_scale: float = 0.0, return_lse: bool = False, precision: str = "auto",
                          lse_layout: int = LSE_NATURAL):
    """q8: row-major fp8 [B,Hq,Sq,D]; k_frag / v_frag: fragment-layout buffers for [B,Hkv,Skv,D].
    v_frag may instead be the ORIGINAL 16-bit value tensor, row-major bf16 / fp16 [B,Hkv,Skv,D] (scale_v = None): the call then runs
    the reference kernel's own P.V numerics -- 16-bit P on the un-quantised V (tk/attention.py:72,286,318) -- for every row.
    return_lse: also the log-sum-exp rows; LSE_REFERENCE gives the reference-defined strided view (include/qattn.h)."""
    _require(q8.is_cuda and q8.dim() == 4, "fp8_attention_forward needs a 4-D device query")
    q8 = q8.contiguous()
    B, Hq, Sq, D = q8.shape
    L = lib()
    mode = _scale_mode(scaling)
    v_is_16 = v_frag.dtype in (torch.bfloat16, torch.float16)
    if v_is_16:
        _require(tuple(v_frag.shape) == (B, Hkv, Skv, D) and scale_v is None and out_dtype == v_frag.dtype,
                 f"a 16-bit value tensor must be [B,Hkv,Skv,D] = {(B, Hkv, Skv, D)} with scale_v=None and out_dtype = its dtype")
        v_frag = v_frag.contiguous()
    _check_scales(scale_q, scale_k, scale_v, mode, B, Hq, Hkv, Sq, Skv, q8.device)
    _require(k_frag.numel() * k_frag.element_size() >= L.qattn_fp8_tensor_bytes(LAYOUT_KFRAG, B, Hkv, Skv, D)
             and (v_is_16 or v_frag.numel() * v_frag.element_size() >= L.qattn_fp8_tensor_bytes(LAYOUT_VFRAG, B, Hkv, Skv, D)),
             "k_frag / v_frag are smaller than the fragment layouts of [B,Hkv,Skv,D]")
    with torch.cuda.device(q8.device):
        out = torch.empty((B, Hq, Sq, D), dtype=out_dtype, device=q8.device)
        lse = None
        if return_lse:
            stride = L.qattn_lse_row_stride(Sq, lse_layout)
            lse = torch.empty((B, Hq, stride), dtype=torch.float32, device=q8.device)
        ws_bytes = L.qattn_attention_workspace_bytes(B, Hq, Sq)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=q8.device)
        rc = L.qattn_fp8_attention_forward(
            q8.data_ptr(), k_frag.data_ptr(), v_frag.data_ptr(), out.data_ptr(), _ptr(lse),
            scale_q.contiguous().data_ptr(), scale_k.contiguous().data_ptr(),
            _ptr(scale_v.contiguous() if scale_v is not None else None),
            B, Hq, Hkv, Sq, Skv, D, fmt_of(q8.dtype), fmt_of(v_frag.dtype if v_is_16 else q8.dtype), fmt_of(out_dtype), mode, int(is_causal),
            float(sm_scale), _precision(precision), lse_layout, ws.data_ptr(), ws_bytes, _stream(q8))
    _check(rc, "qattn_fp8_attention_forward")
    if return_lse:
        return out, lse[..., :Sq]  # [B,Hq,Sq] view; strides (Hq*ld, ld, 1) with ld = the padded row for LSE_REFERENCE
    return out


def describe_path(entry: str, D: int, in_dtype=torch.bfloat16, scaling: str = "head-wise", Skv: int = 4096, want_lse: bool = False) -> dict:
    """qattn_describe_path as a dict of names (host-only: no device call)."""
    _require(entry in ENTRY, f"unknown entry {entry!r} (expected one of {sorted(ENTRY)})")
    d = PathDesc()
    _check(lib().qattn_describe_path(ENTRY[entry], D, fmt_of(in_dtype), _scale_mode(scaling), Skv, int(want_lse), ctypes.byref(d)), "qattn_describe_path")
    return {n: PATH_DESC_NAMES[n][getattr(d, n)] for n, _ in PathDesc._fields_}


def fp8_attention_forward_rowmajor(q8: torch.Tensor, k8: torch.Tensor, v16: torch.Tensor, scale_q: torch.Tensor, scale_k: torch.Tensor, *,
                                   is_causal: bool, pv_16bit: bool = False, sm_scale: float = 0.0, precision: str = "auto",
                                   return_lse: bool = False, lse_layout: int = LSE_NATURAL):
    """The pybind function's contract in one C call (qattn_fp8_attention_forward_rowmajor; tk/attention.py:357-360): row-major fp8 q8
    [B,Hq,Sq,D] / k8 [B,Hkv,Skv,D], 16-bit v16, fp32 scales -> out in v16's dtype.  pv_16bit: V and P stay 16-bit (the reference kernel's
    numerics) instead of both GEMMs on FP8 MFMA."""
    _require(q8.is_cuda and q8.dim() == 4 and k8.dim() == 4 and v16.dim() == 4, "q8, k8 and v16 must be 4-D device tensors")
    _require(q8.dtype in (torch.float8_e4m3fn, torch.float8_e5m2) and k8.dtype == q8.dtype, "q8 / k8 must share an fp8 dtype")
    _require(v16.dtype in (torch.bfloat16, torch.float16), "v16 must be bf16 or fp16")
    q8, k8, v16 = q8.contiguous(), k8.contiguous(), v16.contiguous()
    B, Hq, Sq, D = q8.shape
    _require(k8.shape == v16.shape and k8.shape[0] == B and k8.shape[3] == D, "k8 / v16 shapes do not match q8")
    Hkv, Skv = k8.shape[1], k8.shape[2]
    _require(Hkv > 0 and Hq % Hkv == 0, f"Hq={Hq} is not a multiple of Hkv={Hkv}")
    mode = SCALE_HEAD if scale_q.dim() == 2 else SCALE_TOKEN
    _check_scales(scale_q, scale_k, None, mode, B, Hq, Hkv, Sq, Skv, q8.device)
    L = lib()
    with torch.cuda.device(q8.device):
        out = torch.empty((B, Hq, Sq, D), dtype=v16.dtype, device=q8.device)
        lse = torch.empty((B, Hq, L.qattn_lse_row_stride(Sq, lse_layout)), dtype=torch.float32, device=q8.device) if return_lse else None
        ws_bytes = L.qattn_fp8_attention_rowmajor_workspace_bytes(B, Hq, Hkv, Sq, Skv, D)
        ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=q8.device)
        rc = L.qattn_fp8_attention_forward_rowmajor(
            q8.data_ptr(), k8.data_ptr(), v16.data_ptr(), out.data_ptr(), _ptr(lse), scale_q.contiguous().data_ptr(),
            scale_k.contiguous().data_ptr(), B, Hq, Hkv, Sq, Skv, D, fmt_of(q8.dtype), fmt_of(v16.dtype),
            fmt_of(v16.dtype) if pv_16bit else fmt_of(q8.dtype), mode, int(is_causal), float(sm_scale), _precision(precision), lse_layout,
            ws.data_ptr(), ws_bytes, _stream(q8))
    _check(rc, "qattn_fp8_attention_forward_rowmajor")
    return (out, lse[..., :Sq]) if return_lse else out


def pack16(x: torch.Tensor, layout: int) -> torch.Tensor:
    """row-major bf16/fp16 [B,H,S,D] -> flat uint8 buffer in K16FRAG / V16FRAG layout."""
    _require(x.is_cuda and x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float16), "pack16 needs a 4-D bf16/fp16 device tensor")
    x = x if _strided_ok(x) else x.contiguous()     # (a strided view with head_dim innermost is read as it is: include/qattn_strided.h)
    B, H, S, D = x.shape
    L = lib()
    with torch.cuda.device(x.device):
        out = torch.empty((L.qattn_16bit_tensor_bytes(layout, B, H, S, D),), dtype=torch.uint8, device=x.device)
        rc = L.qattn_pack16_strided(x.data_ptr(), _strides3(x), out.data_ptr(), B, H, S, D, layout, _stream(x))
    _check(rc, "qattn_pack16_strided")
    return out


def attention_forward_16(q: torch.Tensor, k_frag: torch.Tensor, v_frag: torch.Tensor, *, Hkv: int, Skv: int,
                         is_causal: bool, sm_scale: float = 0.0, return_lse: bool = False, fast_exp: bool = False,
                         output_layout: str = "contiguous"):
    """q: row-major bf16/fp16 [B,Hq,Sq,D]; k_frag / v_frag: K16FRAG / V16FRAG buffers for [B,Hkv,Skv,D]."""
    _require(q.is_cuda and q.dim() == 4, "attention_forward_16 needs a 4-D device query")
    q = q if _strided_ok(q) else q.contiguous()
    B, Hq, Sq, D = q.shape
    L = lib()
    _require(k_frag.numel() * k_frag.element_size() >= L.qattn_16bit_tensor_bytes(LAYOUT_K16FRAG, B, Hkv, Skv, D)
             and v_frag.numel() * v_frag.element_size() >= L.qattn_16bit_tensor_bytes(LAYOUT_V16FRAG, B, Hkv, Skv, D),
             "k_frag / v_frag are smaller than the fragment layouts of [B,Hkv,Skv,D]")
    with torch.cuda.device(q.device):
        out = empty_output(q, q.dtype, output_layout)   # dense whatever the strides of q, unless the caller asks for q's layout
        strides = None
        if not (q.is_contiguous() and out.is_contiguous()):
            dense = (Hq * Sq * D, Sq * D, D)
            strides = (ctypes.c_longlong * 6)(*[t.stride(i) if t.shape[i] > 1 else dense[i] for t in (q, out) for i in (0, 1, 2)])
        lse = torch.empty((B, Hq, Sq), dtype=torch.float32, device=q.device) if return_lse else None
        rc = L.qattn_attention_forward_16_strided(q.data_ptr(), strides, k_frag.data_ptr(), v_frag.data_ptr(), out.data_ptr(), _ptr(lse),
                                                  B, Hq, Hkv, Sq, Skv, D, fmt_of(q.dtype), int(is_causal), float(sm_scale),
                                                  int(fast_exp), _stream(q))
    _check(rc, "qattn_attention_forward_16_strided")
    return (out, lse) if return_lse else out


def _per_head(name, t, B, H, device):
    """producer-side per-head figure: fp32 [B,H] on the tensors' device (qattn_fp8_quant_attention_forward_ex)"""
    if t is None:
        return None
    _require(t.dtype == torch.float32 and t.device == device and tuple(t.shape) == (B, H), f"{name} must be float32 {(B, H)} on {device}")
    return t.contiguous()


def fp8_quant_attention_forward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, is_causal: bool, scaling: str = "head-wise",
                                fp8_dtype=torch.float8_e4m3fn, numerics: str = "compiled", sm_scale: float = 0.0,
                                precision: str = "auto", amax_q: Optional[torch.Tensor] = None,
                                amax_k: Optional[torch.Tensor] = None, amax_v: Optional[torch.Tensor] = None,
                                ssq_q: Optional[torch.Tensor] = None, ssq_k: Optional[torch.Tensor] = None,
                                return_lse: bool = False, lse_layout: int = LSE_NATURAL, return_path: bool = False,
                                output_layout: str = "contiguous"):
    """16-bit q, k, v -> attention output: the quant pre-pass and the attention launch(es) in ONE C call
    (qattn_fp8_quant_attention_forward_ex); the pre-pass skips Q where the attention kernel quantises it itself.
    amax_* / ssq_*: per-head abs-max / sum of squares a producer of q, k, v already has (head-wise scaling only): the
    abs-max launch then skips those tensors, or is skipped.
    return_lse: also the per-row log-sum-exp [B,Hq,Sq] (the vector the reference defines, tk/attention.py:333-346), written by the same
    launch as the output.  return_path: also the uint8 [B,Hq,Sq] PATH_* code of every row (test / debug output).
    Returns out, or (out, lse), (out, path), (out, lse, path)."""
    B, Hq, Hkv, Sq, Skv, D = _check_qkv(q, k, v)
    # strided views (q = x.view(B, S, H, D).transpose(1, 2), slices of a packed QKV projection, ...) go to the kernels as they are
    # (include/qattn_strided.h); only what the kernels cannot address -- a head_dim that is not innermost and dense, rows off 16 bytes -- is copied
    q, k, v = (t if _strided_ok(t) else t.contiguous() for t in (q, k, v))
    out = empty_output(q, q.dtype, output_layout)
    strides = None
    if not (q.is_contiguous() and k.is_contiguous() and v.is_contiguous() and out.is_contiguous()):
        dense = lambda t: (t.shape[1] * t.shape[2] * D, t.shape[2] * D, D)     # (a dimension of size 1 has no stride of its own)
        strides = (ctypes.c_longlong * 12)(*[t.stride(i) if t.shape[i] > 1 else dense(t)[i] for t in (q, k, v, out) for i in (0, 1, 2)])
    L = lib()
    mode = _scale_mode(scaling)
    dev = q.device
    given = [x is not None for x in (amax_q, amax_k, amax_v, ssq_q, ssq_k)]
    _require(not any(given) or mode == SCALE_HEAD, "amax_* / ssq_* are per-head figures: head-wise scaling only")
    _require((ssq_q is None) == (ssq_k is None), "ssq_q and ssq_k must be both provided or both not provided")
    amax_q, ssq_q = _per_head("amax_q", amax_q, B, Hq, dev), _per_head("ssq_q", ssq_q, B, Hq, dev)
    amax_k, amax_v, ssq_k = (_per_head(n, t, B, Hkv, dev) for n, t in (("amax_k", amax_k), ("amax_v", amax_v), ("ssq_k", ssq_k)))
    with torch.cuda.device(dev):
        q8 = torch.empty((B, Hq, Sq, D), dtype=torch.uint8, device=dev)
        kf = torch.empty((L.qattn_fp8_tensor_bytes(LAYOUT_KFRAG, B, Hkv, Skv, D),), dtype=torch.uint8, device=dev)
        vf = torch.empty((L.qattn_fp8_tensor_bytes(LAYOUT_VFRAG, B, Hkv, Skv, D),), dtype=torch.uint8, device=dev)
        sq = torch.empty((B, Hq) if mode == SCALE_HEAD else (B, Hq, Sq), dtype=torch.float32, device=dev)
        sk = torch.empty((B, Hkv) if mode == SCALE_HEAD else (B, Hkv, Skv), dtype=torch.float32, device=dev)
        sv = torch.empty((B, Hkv), dtype=torch.float32, device=dev)
        ws_bytes = L.qattn_fp8_quant_attention_workspace_bytes(B, Hq, Hkv, Sq)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        lse = torch.empty((B, Hq, L.qattn_lse_row_stride(Sq, lse_layout)), dtype=torch.float32, device=dev) if return_lse else None
        path = torch.empty((B, Hq, Sq), dtype=torch.uint8, device=dev) if return_path else None
        rc = L.qattn_fp8_quant_attention_forward_strided(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), strides, fmt_of(q.dtype), out.data_ptr(), q8.data_ptr(), kf.data_ptr(),
            vf.data_ptr(), sq.data_ptr(), sk.data_ptr(), sv.data_ptr(), _ptr(amax_q), _ptr(amax_k), _ptr(amax_v), _ptr(ssq_q),
            _ptr(ssq_k), B, Hq, Hkv, Sq, Skv, D, fmt_of(fp8_dtype), mode,
            _numerics(numerics), int(is_causal), float(sm_scale), _precision(precision), _ptr(lse), lse_layout, _ptr(path),
            ws.data_ptr(), ws_bytes, _stream(q))
    _check(rc, "qattn_fp8_quant_attention_forward_strided")
    if not (return_lse or return_path):
        return out
    return (out,) + ((lse[..., :Sq],) if return_lse else ()) + ((path,) if return_path else ())


def measure_attention_clock(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, is_causal: bool = False, precision: str = "auto",
                            fp8_dtype=torch.float8_e4m3fn, calls: int = 50):
    """Measurement aid: run the fused step `calls` times back to back on the stamped instantiation of the attention kernel
    (qattn_fp8_quant_attention_forward_stamped) and return (median in-kernel clock in GHz over the waves of the LAST call, median
    sweep cycles per wave, output of the last call).  D = 128, bf16, head-wise, e4m3 only."""
    B, Hq, Hkv, Sq, Skv, D = _check_qkv(q, k, v)
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    L = lib()
    dev = q.device
    with torch.cuda.device(dev):
        out = torch.empty_like(q)
        q8 = torch.empty((B, Hq, Sq, D), dtype=torch.uint8, device=dev)
        kf = torch.empty((L.qattn_fp8_tensor_bytes(LAYOUT_KFRAG, B, Hkv, Skv, D),), dtype=torch.uint8, device=dev)
        vf = torch.empty((L.qattn_fp8_tensor_bytes(LAYOUT_VFRAG, B, Hkv, Skv, D),), dtype=torch.uint8, device=dev)
        sq = torch.empty((B, Hq), dtype=torch.float32, device=dev)
        sk, sv = (torch.empty((B, Hkv), dtype=torch.float32, device=dev) for _ in range(2))
        ws_bytes = L.qattn_fp8_quant_attention_workspace_bytes(B, Hq, Hkv, Sq)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        st_bytes = L.qattn_attention_stamp_bytes(B, Hq, Sq)
        stamps = torch.zeros((st_bytes // 8,), dtype=torch.int64, device=dev)
        for _ in range(calls):
            rc = L.qattn_fp8_quant_attention_forward_stamped(
                q.data_ptr(), k.data_ptr(), v.data_ptr(), fmt_of(q.dtype), out.data_ptr(), q8.data_ptr(), kf.data_ptr(), vf.data_ptr(),
                sq.data_ptr(), sk.data_ptr(), sv.data_ptr(), B, Hq, Hkv, Sq, Skv, D, fmt_of(fp8_dtype), SCALE_HEAD, NUMERICS["compiled"],
                int(is_causal), 0.0, _precision(precision), ws.data_ptr(), ws_bytes, stamps.data_ptr(), st_bytes, _stream(q))
            _check(rc, "qattn_fp8_quant_attention_forward_stamped")
        torch.cuda.synchronize(dev)
    s = stamps.view(-1, 2).double().cpu()
    s = s[s[:, 1] > 0]
    clock = (s[:, 0] / s[:, 1] * 0.1)
    return float(clock.median()), float(s[:, 0].median()), out


def measure_mfma_peak(device=None, *, seconds: float = 0.3, iters: int = 20000, constant: bool = False):
    """Measurement aid (qattn_mfma_probe): TFLOP/s and in-kernel clock (GHz) of a bare v_mfma_f32_32x32x64_f8f6f4 loop on random
    e4m3 operand bytes (|x| in [2^-2, 2^2), random sign and mantissa -- the regime of quantised N(0,1) data), two waves per SIMD on
    every CU, launched back to back for `seconds`; the figures are those of the last third of the run (settled clocks)."""
    L = lib()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(dev):
        nbytes = L.qattn_mfma_probe_bytes()
        scratch = torch.zeros((nbytes,), dtype=torch.uint8, device=dev)
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        n_op = 2048 * 32
        if constant:
            scratch[:n_op] = 0x38
        else:
            expo = torch.randint(0x28, 0x48, (n_op,), generator=g, device=dev, dtype=torch.int32)
            sign = torch.randint(0, 2, (n_op,), generator=g, device=dev, dtype=torch.int32) << 7
            scratch[:n_op] = (expo | sign).to(torch.uint8)
        fl, nw = ctypes.c_double(0.0), ctypes.c_int(0)
        call = lambda: _check(L.qattn_mfma_probe(scratch.data_ptr(), nbytes, iters, ctypes.byref(fl), ctypes.byref(nw), _stream(scratch)),
                              "qattn_mfma_probe")
        call()
        torch.cuda.synchronize(dev)
        # the lap count bounds GPU time, not host time: launches are asynchronous (6 ms each), and a loop bounded by the host clock
        # queued 5500 of them -- 33 s of dense matrix work in front of every later measurement
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call()
        e1.record()
        torch.cuda.synchronize(dev)
        n_laps = max(3, min(200, int(math.ceil(seconds * 1e3 / max(e0.elapsed_time(e1), 1e-3)))))
        laps = []
        for _ in range(n_laps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call()
            e1.record()
            laps.append((e0, e1))
        torch.cuda.synchronize(dev)
        ms = sorted(a.elapsed_time(b) for a, b in laps[len(laps) * 2 // 3:])
        med = ms[len(ms) // 2]
        st = scratch[n_op:n_op + 16 * nw.value].view(torch.int64).view(-1, 2).double().cpu()
        st = st[st[:, 1] > 0]
        clock = float((st[:, 0] / st[:, 1] * 0.1).median())
    return {"TFLOPs": fl.value / (med * 1e-3) / 1e12, "in_kernel_clock_ghz": clock, "ms_per_launch": med, "launches": len(laps),
            "waves": nw.value, "operands": "constant 1.0" if constant else "random e4m3, |x| in [2^-2, 2^2)"}
