// qattn_attn_pv16.hip -- qattn_fp8_attention_forward with a 16-bit V (v_fmt = QATTN_FMT_BF16 / QATTN_FMT_FP16): the reference kernel's
// own numerics (FP8 QK^T, 16-bit P and V: src/quantum_attn/tk/attention.py:72,286,318) for whole tensors.  Every 256-row query block
// runs pv16_block_pass (qattn_pv16.h), D = 64 / 128 / 256; the bf16 fused step reaches the same pass from inside the D = 128 kernel for
// its early rows, every other fused path through a launch of this kernel over the early blocks.
#include "qattn_pv16.h"

namespace qattn {

// (D = 256: 128 registers of O^T per lane -- one workgroup per CU, which its 144 KiB ring asks for anyway)
template <int D, int QK_FMT, int V16_FMT, bool CAUSAL, bool TOKEN>
__global__ __launch_bounds__(kThreads, D == 256 ? 1 : 2) void attn_pv16_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    pv16_block_pass<D, kWaves, QK_FMT, V16_FMT, CAUSAL, TOKEN, false>(p, smem, (int)threadIdx.x, (int)blockIdx.x, []() { return 0u; }, [](unsigned) {});
}

template <int D, int QK_FMT, int V16_FMT, bool CAUSAL, bool TOKEN>
static int launch_one(const AttnParams& p, hipStream_t st) {
    constexpr int lds = kPv16Slots * (64 * D + 64 * D * 2);
    auto kern = attn_pv16_kernel<D, QK_FMT, V16_FMT, CAUSAL, TOKEN>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return QATTN_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.B * p.Hq * p.nqb)), dim3(kThreads), lds, st, p);
    return QATTN_OK;
}

template <int D, int QK_FMT, int V16_FMT>
static int launch_fmt(const AttnParams& p, int causal, int scale_mode, hipStream_t st) {
    const bool tok = scale_mode == QATTN_SCALE_TOKEN;
    if (causal) return tok ? launch_one<D, QK_FMT, V16_FMT, true, true>(p, st) : launch_one<D, QK_FMT, V16_FMT, true, false>(p, st);
    return tok ? launch_one<D, QK_FMT, V16_FMT, false, true>(p, st) : launch_one<D, QK_FMT, V16_FMT, false, false>(p, st);
}

template <int D>
static int launch_d(const AttnParams& p, int qk_fmt, int v16_fmt, int causal, int scale_mode, hipStream_t st) {
    if (qk_fmt == QATTN_FMT_E4M3)
        return v16_fmt == QATTN_FMT_BF16 ? launch_fmt<D, QATTN_FMT_E4M3, QATTN_FMT_BF16>(p, causal, scale_mode, st)
                                         : launch_fmt<D, QATTN_FMT_E4M3, QATTN_FMT_FP16>(p, causal, scale_mode, st);
    return v16_fmt == QATTN_FMT_BF16 ? launch_fmt<D, QATTN_FMT_E5M2, QATTN_FMT_BF16>(p, causal, scale_mode, st)
                                     : launch_fmt<D, QATTN_FMT_E5M2, QATTN_FMT_FP16>(p, causal, scale_mode, st);
}

int launch_attn_pv16(const AttnParams& pin, int D, int qk_fmt, int v16_fmt, int causal, int scale_mode, hipStream_t st, int n_blocks) {
    AttnParams p = pin;
    if (n_blocks > 0) { p.nqb = n_blocks < p.nqb ? n_blocks : p.nqb; p.risky_lo = p.risky_hi = 0; }   // (the grid and map_block follow nqb)
    if (D == 64) return launch_d<64>(p, qk_fmt, v16_fmt, causal, scale_mode, st);
    if (D == 128) return launch_d<128>(p, qk_fmt, v16_fmt, causal, scale_mode, st);
    if (D == 256) return launch_d<256>(p, qk_fmt, v16_fmt, causal, scale_mode, st);
    return QATTN_ERR_UNSUPPORTED_DIM;
}

}  // namespace qattn
