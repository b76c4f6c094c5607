"""dtype predicates kept for callers of the reference's quantum_attn.utils.types (same two names)."""
import torch

_FP8_DTYPES = frozenset(d for d in (getattr(torch, n, None) for n in ("float8_e4m3fn", "float8_e5m2", "float8_e4m3fnuz", "float8_e5m2fnuz")) if d is not None)


def is_8bit_type(dtype: torch.dtype) -> bool:
    """one byte per element (fp8, int8, uint8, bool ...)"""
    return torch.empty((), dtype=dtype).element_size() == 1


def is_fp8_type(dtype: torch.dtype) -> bool:
    """one of torch's 8-bit floating-point formats"""
    return dtype in _FP8_DTYPES
