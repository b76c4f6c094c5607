// qattn_attn_pv16.hip -- qattn_fp8_attention_forward with a 16-bit V (v_fmt = QATTN_FMT_BF16 / QATTN_FMT_FP16): the reference kernel's
// own numerics (FP8 QK^T, 16-bit P and V: src/quantum_attn/tk/attention.py:72,286,318) for whole tensors.  Every 256-row query block
// runs pv16_block_pass (qattn_pv16.h), D = 64 / 128 / 256; the bf16 fused step reaches the same pass from inside the D = 128 kernel for
// its early rows, every other fused path through a launch of this kernel over the early blocks.
#include "qattn_pv16.h"

namespace qattn {

// (two waves per SIMD = 256 registers: D = 256 holds 128 of O^T per lane; one workgroup per CU, which its 144 KiB ring asks for anyway)
// PP: the two-group loop with a four-stage ring (whole-tensor launches at D = 128, qattn_pv16.h); else the one-group loop, three stages
// Built twice, like the hand-scheduled kernel (qattn_attn.h QATTN_STRIDED16; build.py: qattn_attn_pv16 with -DQATTN_STRIDED16=0,
// qattn_attn_pv16_sv with -DQATTN_PV16_SV): with run-time strides in the same source the token-wise D = 128 instantiation came out with its 32 key-scale
// loads serialised behind s_waitcnt vmcnt(0) -- +40 % on the early rows, +7 % on a token-wise causal step (profiles/r06/ab_strided_token_causal.log).
// SV: the unit's kStrided16, part of the kernel's NAME only.
#ifdef QATTN_PV16_SV
#define QATTN_PV16_ENTRY launch_attn_pv16_sv
static_assert(kStrided16, "the strided-view unit addresses through the strides");
#else
#define QATTN_PV16_ENTRY launch_attn_pv16_dense
static_assert(!kStrided16, "the dense unit is built with -DQATTN_STRIDED16=0 (qattn_attn.h)");
#endif
template <int D, int QK_FMT, int V16_FMT, bool CAUSAL, bool TOKEN, bool PP, int SV = QATTN_STRIDED16>
__global__ __launch_bounds__(kThreads, 2) void attn_pv16_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    pv16_block_pass<D, kWaves, QK_FMT, V16_FMT, CAUSAL, TOKEN, false, PP ? 4 : 3, PP>(p, smem, (int)threadIdx.x, (int)blockIdx.x, []() { return 0u; }, [](unsigned) {});
}

template <int D, int QK_FMT, int V16_FMT, bool CAUSAL, bool TOKEN, bool PP>
static int launch_pp(const AttnParams& p, hipStream_t st) {
    constexpr int lds = (PP ? 4 : 3) * (64 * D + 64 * D * 2);
    auto kern = attn_pv16_kernel<D, QK_FMT, V16_FMT, CAUSAL, TOKEN, PP>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return QATTN_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.B * p.Hq * p.nqb)), dim3(kThreads), lds, st, p);
    return QATTN_OK;
}
template <int D, int QK_FMT, int V16_FMT, bool CAUSAL, bool TOKEN>
static int launch_one(const AttnParams& p, hipStream_t st) {
    if constexpr (D == 128) {
        if (p.total_blocks == 0) return launch_pp<D, QK_FMT, V16_FMT, CAUSAL, TOKEN, true>(p, st);   // (launch_attn_pv16: 0 = every block of every head)
    }
    return launch_pp<D, QK_FMT, V16_FMT, CAUSAL, TOKEN, false>(p, st);
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

int QATTN_PV16_ENTRY(const AttnParams& pin, int D, int qk_fmt, int v16_fmt, int causal, int scale_mode, hipStream_t st, int n_blocks) {
    AttnParams p = pin;
    p.total_blocks = 0;   // (here: "a whole-tensor launch", read by launch_one)
    if (n_blocks > 0) { p.nqb = n_blocks < p.nqb ? n_blocks : p.nqb; p.risky_lo = p.risky_hi = 0; p.tail_lo = 0; p.total_blocks = 1; }   // (the grid and map_block follow nqb)
    if (D == 64) return launch_d<64>(p, qk_fmt, v16_fmt, causal, scale_mode, st);
    if (D == 128) return launch_d<128>(p, qk_fmt, v16_fmt, causal, scale_mode, st);
    if (D == 256) return launch_d<256>(p, qk_fmt, v16_fmt, causal, scale_mode, st);
    return QATTN_ERR_UNSUPPORTED_DIM;
}

}  // namespace qattn
