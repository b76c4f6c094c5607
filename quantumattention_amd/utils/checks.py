"""Device / version predicates.  Mirrors src/quantum_attn/utils/checks.py, with the reference's NVIDIA gate
(`is_nvidia_cuda` + `cuda_capability_compare("ge", 9, 0)`, checks.py:57-64, nn.py:214) replaced by a gfx950 gate."""
import functools
import importlib

import torch


@torch.compiler.assume_constant_result
def config_value(dotted_name: str):
    """`quantumattention_amd.config.<dotted_name>` as a trace-time constant: dynamo cannot follow the config module's
    __getattr__, so flag reads inside traced code go through this helper (role of the reference's checks.py:9-15)."""
    from functools import reduce

    return reduce(getattr, dotted_name.split("."), importlib.import_module("quantumattention_amd.config"))


def is_amd_rocm() -> bool:
    return torch.version.hip is not None and torch.cuda.is_available()


@functools.lru_cache(maxsize=None)
def _arch_name(index: int) -> str:
    return torch.cuda.get_device_properties(index).gcnArchName


def is_gfx950(device=None) -> bool:
    """True when `device` is an MI355X-class (gfx950 / CDNA4) GPU under PyTorch-ROCm."""
    if not is_amd_rocm():
        return False
    if device is None:
        index = torch.cuda.current_device()
    else:
        device = torch.device(device)
        index = device.index if device.index is not None else torch.cuda.current_device()
    return "gfx950" in _arch_name(index)


def fused_step_scales_v_per_head(D: int, in_dtype, scaling: str, Skv: int) -> bool:
    """Mirror of qattn_fp8_quant_attention_forward's choice (csrc/qattn_api.hip quant_attention_impl, `v_block`): with head-wise scales
    V is quantised per 64-key chunk -- no abs-max of V is read -- on every head dim and for both 16-bit input types (the D = 128 kernel
    quantises Q itself and takes the chunk scales; the templated kernel at D = 64 / 256 takes them too) as long as a head has at most 256
    chunks; only beyond that, and with token-wise scales, does V get one scale per head, i.e. is its per-head abs-max needed.
    (`in_dtype`: no longer part of the rule -- until round 4 fp16 inputs at D = 128 kept the per-head V.)"""
    block = scaling in ("head", "head-wise") and (Skv + 63) // 64 <= 256
    return not block
