"""Flags, same mechanism as src/quantum_attn/config.py (env vars read at import + torch's config module so that
`config.patch({...})` works, config.py:34-41).  Inductor/Triton-only flags of the reference are not reproduced."""
import os  # noqa: C101
import sys

_save_config_ignore = set()


class dynamo:
    # kept for interface compatibility (config.py:14-17); this build never calls torch.compile on the hot path
    dynamic = os.getenv("QUANTUM_ATTN_DYNAMIC") == "1"
    mode = os.getenv("QUANTUM_ATTN_MODE", "default")


class attention:
    skip_supported_check = os.getenv("QUANTUM_ATTN_SKIP_SUPPORTED_CHECK") == "1"
    force_eager_fallback = os.getenv("QUANTUM_ATTN_FORCE_EAGER_FALLBACK") == "1"

    # replaces enable_tk_tma_kernel (config.py:30): gate of the hand-written gfx950 HIP kernel
    enable_hip_kernel = os.getenv("QUANTUM_ATTN_ENABLE_HIP_KERNEL", "1") == "1"

    # build extensions: fp8 format of the quantised operands ("e4m3" = reference, "e5m2") and quantiser numerics
    fp8_format = os.getenv("QUANTUM_ATTN_FP8_FORMAT", "e4m3")
    quant_numerics = os.getenv("QUANTUM_ATTN_QUANT_NUMERICS", "compiled")

    # how the probabilities P enter the second fp8 GEMM:
    #   "auto"     one-term e4m3 P where the softmax row is spread over enough keys; the rows of a 256-row query block that turn
    #              out to be peaked (few keys carry the weight) are recomputed with more precision: two-term (hi + lo) fp8 P, or --
    #              fused step, head_dim 128, head-wise -- 16-bit P on the original 16-bit V (the reference kernel's numerics)
    #   "fast"     one-term P everywhere (what other fp8 attention kernels do); error grows with the sharpness of the rows
    #   "accurate" the precise pass everywhere (about bf16-P accuracy, ~1.5x the matrix work)
    precision = os.getenv("QUANTUM_ATTN_PRECISION", "auto")

    # the P.V product of `fp8_attention_forward` on pre-quantised query / key (the reference's op contract: value arrives in 16 bit):
    #   "fp8"    (default) value is quantised to fp8 per head and both GEMMs run on FP8 MFMA (north_star)
    #   "16bit"  the reference kernel's own numerics: 16-bit P on the un-quantised value, bf16 / fp16 MFMA (head_dim 64 / 128 / 256;
    #            about 1.5x the time).  Independently of this switch the fused step `fp8_attn_func(16-bit q, k, v)` attends the original
    #            16-bit V for the query blocks that see fewer than 1024 keys (early causal rows, short sequences).
    #   Read PER CALL inside the op body (ops.fp8_attention_forward), unlike precision / fp8_format / quant_numerics, which the fused op
    #   takes as arguments and a compiled graph therefore bakes in at trace time: a compiled region follows a later change of this flag.
    pv_precision = os.getenv("QUANTUM_ATTN_PV_PRECISION", "fp8")

    # torch.compile: trace the per-head abs-max (and sums of squares) of query / key / value into the caller's graph as aten reductions, so
    # that Inductor fuses them into the kernel that produced the tensors and the quant pre-pass skips its abs-max launch -- what the
    # reference gets from inlining its quantiser into the compiled region (nn.py:410-418, 484-501).  Eager calls are not affected.
    inline_abs_max_under_compile = os.getenv("QUANTUM_ATTN_INLINE_ABS_MAX", "1") == "1"

    # memory layout of the output: "contiguous" (default) = a fresh dense [B,H,S,D] tensor, as the reference's launcher allocates it
    # (tk/attention.py:434-437); "like_query" = the layout of the query -- for q = x.view(B, S, H, D).transpose(1, 2) the result is the
    # transposed view of a dense [B,S,H,D] tensor (what torch's own flash SDPA returns), so that the caller's
    # `out.transpose(1, 2).reshape(B, S, H * D)` before its output projection is a view instead of a copy.  Same values either way.
    output_layout = os.getenv("QUANTUM_ATTN_OUTPUT_LAYOUT", "contiguous")

    # 16-bit sibling path (attn_func): False = exact v_exp_f32 softmax (default); True = the linear-mantissa 2^x
    # approximation for rows that see >= 1024 keys (+9 % speed, 1.8 % rms error in P: fine for flat rows only)
    fast_exp16 = os.getenv("QUANTUM_ATTN_FAST_EXP16") == "1"


from torch.utils._config_module import install_config_module  # noqa: E402

# adds patch, save_config, etc
install_config_module(sys.modules[__name__])
