"""ctypes binding of libqattn_hip.so (include/qattn.h).  No CPU fallback: a missing library is a hard error."""
import ctypes
import os
from typing import Optional, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libqattn_hip.so")

FMT_E4M3, FMT_E5M2, FMT_BF16, FMT_FP16 = 0, 1, 2, 3
SCALE_HEAD, SCALE_TOKEN = 0, 1
LAYOUT_ROWMAJOR, LAYOUT_KFRAG, LAYOUT_VFRAG, LAYOUT_K16FRAG, LAYOUT_V16FRAG = 0, 1, 2, 3, 4
NUMERICS = {"compiled": 0, "eager": 1}
ABI_VERSION = 3

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
    "qattn_debug_last_attention_ms",
)

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
    L = ctypes.CDLL(LIB_PATH)
    vp, i, f, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
    L.qattn_abi_version.restype = i
    L.qattn_strerror.restype = ctypes.c_char_p
    L.qattn_strerror.argtypes = [i]
    L.qattn_check_device.restype = i
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
    L.qattn_fp8_attention_forward.restype = i
    L.qattn_fp8_attention_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, i, i, f, vp]
    L.qattn_16bit_tensor_bytes.restype = sz
    L.qattn_16bit_tensor_bytes.argtypes = [i, i, i, i, i]
    L.qattn_pack16.restype = i
    L.qattn_pack16.argtypes = [vp, vp, i, i, i, i, i, vp]
    L.qattn_attention_forward_16.restype = i
    L.qattn_attention_forward_16.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, f, vp]
    L.qattn_debug_last_attention_ms.restype = ctypes.c_float
    L.qattn_fp8_quant_attention_forward.restype = i
    L.qattn_fp8_quant_attention_forward.argtypes = [vp, vp, vp, i, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, i, f, vp, sz, vp]
    if L.qattn_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libqattn_hip.so ABI {L.qattn_abi_version()} != expected {ABI_VERSION}; rebuild it")
    _lib = L
    return L


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
    assert x.is_cuda and x.dim() == 4
    x = x.contiguous()
    B, H, S, D = x.shape
    L = lib()
    mode = SCALE_HEAD if scaling == "head-wise" else SCALE_TOKEN
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
                               fmt_of(fp8_dtype), mode, NUMERICS[numerics], layout, ws.data_ptr(), ws_bytes,
                               _stream(x))
    _check(rc, "qattn_quant_fp8")
    return out, scale


def quant_qkv_fp8(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, scaling: str = "head-wise",
                  fp8_dtype=torch.float8_e4m3fn, numerics: str = "compiled"):
    """Fused pre-pass: (q8 row-major fp8, k_frag, v_frag, scale_q, scale_k, scale_v) in two launches."""
    assert q.is_cuda and q.dim() == 4 and k.shape == v.shape and q.dtype == k.dtype == v.dtype
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    B, Hq, Sq, D = q.shape
    _, Hkv, Skv, _ = k.shape
    L = lib()
    mode = SCALE_HEAD if scaling == "head-wise" else SCALE_TOKEN
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
                                   B, Hq, Hkv, Sq, Skv, D, fmt_of(fp8_dtype), mode, NUMERICS[numerics],
                                   ws.data_ptr(), ws_bytes, _stream(q))
    _check(rc, "qattn_quant_qkv_fp8")
    return q8, kf, vf, sq, sk, sv


def pack_fp8(x8: torch.Tensor, layout: int) -> torch.Tensor:
    """row-major fp8 [B,H,S,D] -> flat uint8 buffer in KFRAG / VFRAG layout."""
    assert x8.is_cuda and x8.dim() == 4 and x8.dtype.itemsize == 1
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
                          sm_scale: float = 0.0, return_lse: bool = False):
    """q8: row-major fp8 [B,Hq,Sq,D]; k_frag / v_frag: fragment-layout buffers for [B,Hkv,Skv,D]."""
    assert q8.is_cuda and q8.dim() == 4
    q8 = q8.contiguous()
    B, Hq, Sq, D = q8.shape
    L = lib()
    mode = SCALE_HEAD if scaling == "head-wise" else SCALE_TOKEN
    with torch.cuda.device(q8.device):
        out = torch.empty((B, Hq, Sq, D), dtype=out_dtype, device=q8.device)
        lse = torch.empty((B, Hq, Sq), dtype=torch.float32, device=q8.device) if return_lse else None
        rc = L.qattn_fp8_attention_forward(
            q8.data_ptr(), k_frag.data_ptr(), v_frag.data_ptr(), out.data_ptr(), _ptr(lse),
            scale_q.contiguous().data_ptr(), scale_k.contiguous().data_ptr(),
            _ptr(scale_v.contiguous() if scale_v is not None else None),
            B, Hq, Hkv, Sq, Skv, D, fmt_of(q8.dtype), fmt_of(q8.dtype), fmt_of(out_dtype), mode, int(is_causal),
            float(sm_scale), _stream(q8))
    _check(rc, "qattn_fp8_attention_forward")
    return (out, lse) if return_lse else out


def pack16(x: torch.Tensor, layout: int) -> torch.Tensor:
    """row-major bf16/fp16 [B,H,S,D] -> flat uint8 buffer in K16FRAG / V16FRAG layout."""
    assert x.is_cuda and x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float16)
    x = x.contiguous()
    B, H, S, D = x.shape
    L = lib()
    with torch.cuda.device(x.device):
        out = torch.empty((L.qattn_16bit_tensor_bytes(layout, B, H, S, D),), dtype=torch.uint8, device=x.device)
        rc = L.qattn_pack16(x.data_ptr(), out.data_ptr(), B, H, S, D, layout, _stream(x))
    _check(rc, "qattn_pack16")
    return out


def attention_forward_16(q: torch.Tensor, k_frag: torch.Tensor, v_frag: torch.Tensor, *, Hkv: int, Skv: int,
                         is_causal: bool, sm_scale: float = 0.0, return_lse: bool = False):
    """q: row-major bf16/fp16 [B,Hq,Sq,D]; k_frag / v_frag: K16FRAG / V16FRAG buffers for [B,Hkv,Skv,D]."""
    assert q.is_cuda and q.dim() == 4
    q = q.contiguous()
    B, Hq, Sq, D = q.shape
    L = lib()
    with torch.cuda.device(q.device):
        out = torch.empty_like(q)
        lse = torch.empty((B, Hq, Sq), dtype=torch.float32, device=q.device) if return_lse else None
        rc = L.qattn_attention_forward_16(q.data_ptr(), k_frag.data_ptr(), v_frag.data_ptr(), out.data_ptr(), _ptr(lse),
                                          B, Hq, Hkv, Sq, Skv, D, fmt_of(q.dtype), int(is_causal), float(sm_scale),
                                          _stream(q))
    _check(rc, "qattn_attention_forward_16")
    return (out, lse) if return_lse else out


def fp8_quant_attention_forward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, is_causal: bool, scaling: str = "head-wise",
                                fp8_dtype=torch.float8_e4m3fn, numerics: str = "compiled", sm_scale: float = 0.0) -> torch.Tensor:
    """16-bit q, k, v -> attention output: the quant pre-pass and the attention launch(es) in ONE C call
    (qattn_fp8_quant_attention_forward); the pre-pass skips Q where the attention kernel quantises it itself."""
    assert q.is_cuda and q.dim() == 4 and k.shape == v.shape and q.dtype == k.dtype == v.dtype
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    B, Hq, Sq, D = q.shape
    _, Hkv, Skv, _ = k.shape
    L = lib()
    mode = SCALE_HEAD if scaling == "head-wise" else SCALE_TOKEN
    dev = q.device
    with torch.cuda.device(dev):
        out = torch.empty_like(q)
        q8 = torch.empty((B, Hq, Sq, D), dtype=torch.uint8, device=dev)
        kf = torch.empty((L.qattn_fp8_tensor_bytes(LAYOUT_KFRAG, B, Hkv, Skv, D),), dtype=torch.uint8, device=dev)
        vf = torch.empty((L.qattn_fp8_tensor_bytes(LAYOUT_VFRAG, B, Hkv, Skv, D),), dtype=torch.uint8, device=dev)
        sq = torch.empty((B, Hq) if mode == SCALE_HEAD else (B, Hq, Sq), dtype=torch.float32, device=dev)
        sk = torch.empty((B, Hkv) if mode == SCALE_HEAD else (B, Hkv, Skv), dtype=torch.float32, device=dev)
        sv = torch.empty((B, Hkv), dtype=torch.float32, device=dev)
        ws_bytes = L.qattn_quant_qkv_workspace_bytes(B, Hq, Hkv)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        rc = L.qattn_fp8_quant_attention_forward(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), fmt_of(q.dtype), out.data_ptr(), q8.data_ptr(), kf.data_ptr(),
            vf.data_ptr(), sq.data_ptr(), sk.data_ptr(), sv.data_ptr(), B, Hq, Hkv, Sq, Skv, D, fmt_of(fp8_dtype), mode,
            NUMERICS[numerics], int(is_causal), float(sm_scale), ws.data_ptr(), ws_bytes, _stream(q))
    _check(rc, "qattn_fp8_quant_attention_forward")
    return out
