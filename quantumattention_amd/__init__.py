"""MI355X-native (gfx950 / CDNA4) FP8 fused attention with the call surface of WaveSpeedAI/QuantumAttention.

Swap `import quantum_attn` for `import quantumattention_amd as quantum_attn`: the exported names are the ones of
src/quantum_attn/__init__.py:23-31.  The hot path runs hand-written HIP kernels through a C ABI
(include/qattn.h, libqattn_hip.so); see DESIGN.md.
"""
import torch  # noqa: F401

from . import config, nn, ops  # noqa: F401
from .quantum_attn_interface import (
    attn_func,
    attn_func_with_fallback,
    dynamically_quantize_fp8,
    fp8_attn_func,
    fp8_attn_func_with_fallback,
    fp8_token_wise_attn_func,
    fp8_token_wise_attn_func_with_fallback,
)

__version__ = "0.1.0"

__all__ = [
    "attn_func",
    "attn_func_with_fallback",
    "dynamically_quantize_fp8",
    "fp8_attn_func",
    "fp8_attn_func_with_fallback",
    "fp8_token_wise_attn_func",
    "fp8_token_wise_attn_func_with_fallback",
]
