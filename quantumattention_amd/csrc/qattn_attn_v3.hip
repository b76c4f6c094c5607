// qattn_attn_v3.hip -- FP8 fused attention forward for gfx950 (MI355X / CDNA4): 4 waves x 64 query rows.
//
// Replaces fwd_attend_ker<D,causal,..> + its launcher (src/quantum_attn/tk/attention.py:97-349, 355-647) behind the
// op quantum_attn::fp8_attention_forward (src/quantum_attn/ops.py:98-121).  Same algorithm and numerics as
// qattn_attn_v2.hip (swapped QK^T on v_mfma_f32_32x32x64_f8f6f4, optimistic exponentiation with a rare fix-up,
// byte-exponential fast path, fragment-layout K/V), laid out for what the v2 ablation showed to be the wall
// (profiles/r01_ablation.md): LDS-array pressure and the 256 architectural VGPRs the VALU can address.
//
//  * workgroup = 4 waves, ONE per SIMD, each wave owns 64 query rows = two 32-row q-blocks.  Every K / V fragment read
//    from LDS feeds TWO back-to-back MFMAs (one per q-block): operand reads drop from 8 waves x 22 to 4 waves x 16
//    ds_read_b128 per 64-key chunk.  Q^T fragments stay in registers.
//  * S^T is SINGLE-buffered (64 VGPRs), staggered at tile level.  Iteration t, 18 MFMA slots:
//        slots  0-3   S(t) tile 0 = K.Q^T (both q-blocks)                 VALU: -
//        slots  4-7   S(t) tile 1                                         VALU: softmax of tile 0 -> P(t) dwords 0-3
//        slots  8-17  O += V(t-1).P(t-1), row sums (ones-row MFMA)        VALU: softmax of tile 1 -> P(t) dwords 4-7, maxes
//    then the rare fix-up check.  Tile 0 is overwritten by the next iteration's slots 0-3 only after its softmax is done;
//    P is double-buffered by iteration parity.  MFMA-only data (O, row sums, Q) can live in AGPRs.
//  * stage(t) = {K(t), V(t-1)} is published one barrier EARLY (written at iteration t-2), so the K tile-0 fragments of
//    chunk t+1 are read at the end of iteration t and the matrix pipe restarts right after the barrier.
//  * K/V staging through registers (global_load_dwordx4 -> ds_write_b128), 3-stage ring, one s_barrier per iteration.
#include <type_traits>

#include "qattn_attn.h"

namespace qattn {

constexpr int kV3Waves = 4;
constexpr int kV3QPerWave = 64;
constexpr int kV3QPerWG = kV3Waves * kV3QPerWave;  // 256
constexpr int kV3Stages = 3;
constexpr int kV3D = 128;

struct StageRegs3 {
    static constexpr int ROUNDS = 2 * 64 * kV3D / (kV3Waves * 64 * 16);  // 4
    v4i r[ROUNDS];
};
__device__ __forceinline__ void stage_load3(StageRegs3& sr, const unsigned char* ksrc, const unsigned char* vsrc, int wave, int lane) {
    constexpr int CH = 64 * kV3D;
#pragma unroll
    for (int r = 0; r < StageRegs3::ROUNDS; r++) {
        const int o = r * (kV3Waves * 1024) + (wave << 10);
        const unsigned char* src = (o < CH ? ksrc + o : vsrc + (o - CH)) + (lane << 4);
        sr.r[r] = *reinterpret_cast<const v4i*>(src);
    }
}
__device__ __forceinline__ void stage_write3(const StageRegs3& sr, unsigned char* lds_stage, int wave, int lane) {
#pragma unroll
    for (int r = 0; r < StageRegs3::ROUNDS; r++)
        *reinterpret_cast<v4i*>(lds_stage + r * (kV3Waves * 1024) + (wave << 10) + (lane << 4)) = sr.r[r];
}

template <bool TWO, bool BYTE>
struct WaveState3 {
    v16f o[2][4];            // O^T accumulators [q-block][32-row block of D]
    v16f s[2][2];            // S^T of the current chunk [q-block][32-key tile] (single-buffered)
    v8i p[2][2];             // P^T (e4m3) ping-pong [t&1][q-block]
    v8i pl[TWO ? 2 : 1][2];  // low term of the two-term split
    v8i qf[2][2];            // Q^T fragments [q-block][k-step]
    v8i kpre[2];             // K fragments (tile 0, k-steps 0,1) of the NEXT chunk, read one iteration ahead
    v16f l16[2];             // BYTE: row sums of the quantised P' (every register holds the full sum)
    float m_run[2];          // running max of the raw scores
    float l_run[2];          // exact mode: this lane's partial row sums
    float c[2];              // scale_q*scale_k*sm_scale*log2(e) per q-block row
    unsigned long long seg[6];  // diagnostic builds (ABL & 16): cycles per segment of the iteration
    unsigned long long tlast;
};

// 4 scores -> exponentials -> one dword of the e4m3 P operand (see qattn_attn_v2.hip for both variants)
template <bool TWO, bool FIRST>
__device__ __forceinline__ void exp_group3(const v16f& sx, int j, float c, float mc, float (&acc)[4], v8i& pv, v8i& plv, int w, int seed) {
    float e[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        e[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[4 * j + i], c, mc));
        acc[i] = FIRST ? e[i] : acc[i] + e[i];
    }
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    int ph = cvt_pk_fp8<QATTN_FMT_E4M3, false>(e[0], e[1], seed);
    ph = cvt_pk_fp8<QATTN_FMT_E4M3, true>(e[2], e[3], ph);
    if (TWO) {
        const float h0 = __builtin_amdgcn_cvt_f32_fp8(ph, 0), h1 = __builtin_amdgcn_cvt_f32_fp8(ph, 1);
        const float h2 = __builtin_amdgcn_cvt_f32_fp8(ph, 2), h3 = __builtin_amdgcn_cvt_f32_fp8(ph, 3);
        int plo = cvt_pk_fp8<QATTN_FMT_E4M3, false>(e[0] - h0, e[1] - h1, ph);
        plo = cvt_pk_fp8<QATTN_FMT_E4M3, true>(e[2] - h2, e[3] - h3, plo);
        asm volatile("" : "+v"(plo));
        plv[w] = plo;
    }
    asm volatile("" : "+v"(ph));
    pv[w] = ph;
}
__device__ __forceinline__ void byte_group3(const v16f& sx, int j, float c8, float off8, v8i& pv, int w, int seed) {
    unsigned b = (unsigned)seed;
#pragma unroll
    for (int i = 0; i < 4; i++) b = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(sx[4 * j + i], c8, off8), i, b);
    asm volatile("" : "+v"(b));
    pv[w] = (int)b;
}

__device__ __forceinline__ float tile_max(const v16f& a, const v16f& b) {
    float mx = fmaxf(fmaxf(a[0], a[1]), a[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, a[r]), a[r + 1]);
    mx = fmaxf(fmaxf(mx, a[15]), b[0]);
#pragma unroll
    for (int r = 1; r < 15; r += 2) mx = fmaxf(fmaxf(mx, b[r]), b[r + 1]);
    return fmaxf(mx, b[15]);
}

// token-wise key scales and the ragged-tail / causal-diagonal mask, in place on ONE finished 32-key tile of S^T
template <bool CAUSAL, bool TOKEN>
__device__ __forceinline__ void prep_tile3(v16f& sx, int tile, const AttnParams& p, int k0, int q0b, int qrow, int hh, const float* skt) {
    if (TOKEN) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int kk = k0 + 32 * tile + 8 * j + 4 * hh;
            float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kk + 3 < p.Skv) w = *reinterpret_cast<const float4*>(skt + kk);
            else { if (kk < p.Skv) w.x = skt[kk]; if (kk + 1 < p.Skv) w.y = skt[kk + 1]; if (kk + 2 < p.Skv) w.z = skt[kk + 2]; }
            sx[4 * j + 0] *= w.x; sx[4 * j + 1] *= w.y; sx[4 * j + 2] *= w.z; sx[4 * j + 3] *= w.w;
        }
    }
    const bool need_mask = (k0 + 64 > p.Skv) || (CAUSAL && k0 + 63 > q0b);  // wave-uniform
    if (__builtin_expect(need_mask, 0)) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int key = k0 + 32 * tile + (r & 3) + 8 * (r >> 2) + 4 * hh;
            const bool dead = key >= p.Skv || (CAUSAL && key > qrow);
            sx[r] = dead ? -INFINITY : sx[r];
        }
    }
}

#define QATTN3_FENCE() __builtin_amdgcn_sched_barrier(0)
#define QATTN3_STAMP(I)                                                                                          \
    do {                                                                                                        \
        if (ABL & 16) {                                                                                         \
            unsigned long long t_;                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
            st.seg[I] += t_ - st.tlast;                                                                         \
            st.tlast = t_;                                                                                      \
        }                                                                                                       \
    } while (0)
// softmax group: q-block B, tile TL, group J of 4 scores of S(t) -> dword 4*TL+J of P(t)
#define QATTN3_SM(FIRST, B, TL, J, MC, SEED)                                                                        \
    do {                                                                                                            \
        if (ABL & 4) break;                                                                                         \
        if (BYTE) byte_group3(st.s[B][TL], J, cx[B], MC[B], pc##B, 4 * (TL) + (J), SEED);                            \
        else exp_group3<TWO, FIRST>(st.s[B][TL], J, cx[B], MC[B], acc##B, pc##B, pcl##B, 4 * (TL) + (J), SEED);      \
    } while (0)
#define QATTN3_PV(B, M, FR)                                                                                         \
    do {                                                                                                            \
        st.o[B][M] = mfma_f8<FMT, QATTN_FMT_E4M3>(FR, pp##B, st.o[B][M]);                                            \
        if (TWO) st.o[B][M] = mfma_f8<FMT, QATTN_FMT_E4M3>(FR, ppl##B, st.o[B][M]);                                  \
    } while (0)

__device__ __forceinline__ float max8(float mx, const v16f& a, int lo) {
#pragma unroll
    for (int r = 0; r < 8; r += 2) mx = fmaxf(fmaxf(mx, a[lo + r]), a[lo + r + 1]);
    return mx;
}

// One iteration (0 <= t < n_w): QK(t), softmax(t), PV(t-1).  PAR = t & 1 selects the P buffers.
//   kbuf  : stage(t)   K part (+ lane offset)        vbuf : stage(t) V part = V(t-1)
//   knext : stage(t+1) K part (already visible)  -> st.kpre for the next iteration
// prep (token-wise key scales / masks) is applied in place on each S tile right after its MFMAs (PREP functor).
template <int FMT, int PAR, bool TWO, bool BYTE, int ABL, typename Prep, typename Stage>
__device__ __forceinline__ void full_step3(WaveState3<TWO, BYTE>& st, const unsigned char* kbuf, const unsigned char* vbuf,
                                           const unsigned char* knext, Prep&& prep, Stage&& stage) {
    // ABL (timing-only ablations): 4 = no softmax VALU, 8 = no LDS fragment reads
    auto LDSF3 = [&](const unsigned char* ptr) -> v8i { if (ABL & 8) return st.qf[0][0]; return lds_read_frag(ptr); };
    constexpr int PL_W = TWO ? PAR : 0, PL_R = TWO ? (PAR ^ 1) : 0;
    v8i& pc0 = st.p[PAR][0]; v8i& pc1 = st.p[PAR][1];                  // P(t) being produced
    v8i& pcl0 = st.pl[PL_W][0]; v8i& pcl1 = st.pl[PL_W][1];
    const v8i& pp0 = st.p[PAR ^ 1][0]; const v8i& pp1 = st.p[PAR ^ 1][1];  // P(t-1) consumed by PV
    const v8i& ppl0 = st.pl[PL_R][0]; const v8i& ppl1 = st.pl[PL_R][1];
    constexpr float SHIFT = BYTE ? kPShiftByte : kPShift, THR = BYTE ? kRescaleThrByte : kRescaleThr;
    float cx[2], mc[2];
#pragma unroll
    for (int b = 0; b < 2; b++) {
        cx[b] = BYTE ? 8.0f * st.c[b] : st.c[b];
        mc[b] = BYTE ? __builtin_fmaf(-8.0f * st.m_run[b], st.c[b], 8.0f * SHIFT + 56.0f + kByteBias) : SHIFT - st.m_run[b] * st.c[b];
    }
    float acc0[4], acc1[4];

    // ---- slots 0-3: S(t) tile 0 (both q-blocks); K(tile 0) fragments were read last iteration.  The VALU has nothing
    // of this chunk to do yet, so the K/V staging of a later chunk (ds_write of the registers loaded last iteration,
    // then the next global loads) is issued here, off the critical path.
#pragma unroll
    for (int r = 0; r < 16; r++) { st.s[0][0][r] = 0.0f; st.s[1][0][r] = 0.0f; st.s[0][1][r] = 0.0f; st.s[1][1][r] = 0.0f; }
    st.s[0][0] = mfma_f8<FMT, FMT>(st.kpre[0], st.qf[0][0], st.s[0][0]);   // slot 0
    v8i kb = LDSF3(kbuf + (2 << 11));                                      // K (tile 1, k-step 0)
    QATTN3_FENCE();
    st.s[1][0] = mfma_f8<FMT, FMT>(st.kpre[0], st.qf[1][0], st.s[1][0]);   // slot 1
    v8i kd = LDSF3(kbuf + (3 << 11));                                      // K (tile 1, k-step 1)
    QATTN3_FENCE();
    st.s[0][0] = mfma_f8<FMT, FMT>(st.kpre[1], st.qf[0][1], st.s[0][0]);   // slot 2
    stage(0);                                                              // ds_write stage(t+2)
    QATTN3_FENCE();
    st.s[1][0] = mfma_f8<FMT, FMT>(st.kpre[1], st.qf[1][1], st.s[1][0]);   // slot 3
    stage(1);                                                              // global loads of stage(t+3)
    v8i v0 = LDSF3(vbuf + (0 << 11));
    QATTN3_FENCE();
    QATTN3_STAMP(1);
    // ---- slots 4-7: S(t) tile 1; from here on one softmax group (4 scores -> 1 dword of P) per slot, two in three slots
    st.s[0][1] = mfma_f8<FMT, FMT>(kb, st.qf[0][0], st.s[0][1]);           // slot 4
    prep(0, 0);
    QATTN3_SM(true, 0, 0, 0, mc, pp0[0]);
    QATTN3_FENCE();
    st.s[1][1] = mfma_f8<FMT, FMT>(kb, st.qf[1][0], st.s[1][1]);           // slot 5
    prep(1, 0);
    v8i v1 = LDSF3(vbuf + (1 << 11));
    QATTN3_SM(true, 1, 0, 0, mc, pp1[0]);
    QATTN3_FENCE();
    st.s[0][1] = mfma_f8<FMT, FMT>(kd, st.qf[0][1], st.s[0][1]);           // slot 6
    QATTN3_SM(false, 0, 0, 1, mc, pc0[0]);
    float mx0 = max8(fmaxf(st.s[0][0][0], st.s[0][0][1]), st.s[0][0], 0);
    QATTN3_FENCE();
    st.s[1][1] = mfma_f8<FMT, FMT>(kd, st.qf[1][1], st.s[1][1]);           // slot 7
    v8i v2 = LDSF3(vbuf + (2 << 11));
    QATTN3_SM(false, 1, 0, 1, mc, pc1[0]);
    float mx1 = max8(fmaxf(st.s[1][0][0], st.s[1][0][1]), st.s[1][0], 0);
    QATTN3_FENCE();
    QATTN3_STAMP(2);
    // ---- slots 8-17: PV(t-1) + row sums
    QATTN3_PV(0, 0, v0);                                                   // slot 8
    QATTN3_SM(false, 0, 0, 2, mc, pc0[1]);
    mx0 = max8(mx0, st.s[0][0], 8);
    QATTN3_FENCE();
    QATTN3_PV(1, 0, v0);                                                   // slot 9
    v8i v3 = LDSF3(vbuf + (3 << 11));
    QATTN3_SM(false, 1, 0, 2, mc, pc1[1]);
    mx1 = max8(mx1, st.s[1][0], 8);
    QATTN3_FENCE();
    QATTN3_PV(0, 1, v1);                                                   // slot 10
    QATTN3_SM(false, 0, 0, 3, mc, pc0[2]);
    QATTN3_SM(false, 1, 0, 3, mc, pc1[2]);
    QATTN3_FENCE();
    QATTN3_PV(1, 1, v1);                                                   // slot 11
    prep(0, 1);
    st.kpre[0] = LDSF3(knext + (0 << 11));                                 // next chunk's K (tile 0, k-step 0)
    QATTN3_SM(false, 0, 1, 0, mc, pc0[3]);
    mx0 = max8(mx0, st.s[0][1], 0);
    QATTN3_FENCE();
    QATTN3_PV(0, 2, v2);                                                   // slot 12
    prep(1, 1);
    QATTN3_SM(false, 1, 1, 0, mc, pc1[3]);
    mx1 = max8(mx1, st.s[1][1], 0);
    QATTN3_FENCE();
    QATTN3_PV(1, 2, v2);                                                   // slot 13
    st.kpre[1] = LDSF3(knext + (1 << 11));                                 // next chunk's K (tile 0, k-step 1)
    QATTN3_SM(false, 0, 1, 1, mc, pc0[4]);
    mx0 = max8(mx0, st.s[0][1], 8);
    QATTN3_FENCE();
    QATTN3_PV(0, 3, v3);                                                   // slot 14
    QATTN3_SM(false, 1, 1, 1, mc, pc1[4]);
    mx1 = max8(mx1, st.s[1][1], 8);
    QATTN3_FENCE();
    QATTN3_PV(1, 3, v3);                                                   // slot 15
    QATTN3_SM(false, 0, 1, 2, mc, pc0[5]);
    QATTN3_SM(false, 1, 1, 2, mc, pc1[5]);
    QATTN3_FENCE();
    if (BYTE) {
        // slots 16,17: row sums of the quantised P(t-1): ones(32x64) . P^T -> every row = the sum over the 64 keys
        v8i ones;
#pragma unroll
        for (int w = 0; w < 8; w++) ones[w] = FMT == QATTN_FMT_E4M3 ? 0x38383838 : 0x3c3c3c3c;  // 1.0 in e4m3 / e5m2
        st.l16[0] = mfma_f8<FMT, QATTN_FMT_E4M3>(ones, pp0, st.l16[0]);   // slot 16
        QATTN3_SM(false, 0, 1, 3, mc, pc0[6]);
        QATTN3_FENCE();
        st.l16[1] = mfma_f8<FMT, QATTN_FMT_E4M3>(ones, pp1, st.l16[1]);   // slot 17
        QATTN3_SM(false, 1, 1, 3, mc, pc1[6]);
    } else {
        QATTN3_SM(false, 0, 1, 3, mc, pc0[6]);
        QATTN3_SM(false, 1, 1, 3, mc, pc1[6]);
    }
    {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx0), __float_as_uint(mx0), false, false);
        mx0 = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx1), __float_as_uint(mx1), false, false);
        mx1 = fmaxf(__uint_as_float(sx[0]), __uint_as_float(sx[1]));
    }
    float ls0 = BYTE ? 0.0f : (acc0[0] + acc0[1]) + (acc0[2] + acc0[3]);
    float ls1 = BYTE ? 0.0f : (acc1[0] + acc1[1]) + (acc1[2] + acc1[3]);
    QATTN3_FENCE();
    QATTN3_STAMP(3);
    // ---- rare fix-up (always on the first chunk): rescale what is accumulated (O and the row sums hold chunks
    // <= t-1), then redo this chunk's exponentials against the new max
    const bool need0 = (mx0 - st.m_run[0]) * st.c[0] > THR, need1 = (mx1 - st.m_run[1]) * st.c[1] > THR;
    if (__builtin_expect(__any(need0 || need1) != 0, 0)) {
        float mc2[2];
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const float m_new = fmaxf(st.m_run[b], b ? mx1 : mx0);
            const float alpha = __builtin_amdgcn_exp2f((st.m_run[b] - m_new) * st.c[b]);
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int r = 0; r < 16; r++) st.o[b][m][r] *= alpha;
            st.l_run[b] *= alpha;
            if (BYTE) {
#pragma unroll
                for (int r = 0; r < 16; r++) st.l16[b][r] *= alpha;
            }
            st.m_run[b] = m_new;
            mc2[b] = BYTE ? __builtin_fmaf(-8.0f * m_new, st.c[b], 8.0f * SHIFT + 56.0f + kByteBias) : SHIFT - m_new * st.c[b];
        }
        QATTN3_SM(true, 0, 0, 0, mc2, 0);
        QATTN3_SM(true, 1, 0, 0, mc2, 0);
#pragma unroll
        for (int j = 1; j < 4; j++) { QATTN3_SM(false, 0, 0, j, mc2, 0); QATTN3_SM(false, 1, 0, j, mc2, 0); }
#pragma unroll
        for (int j = 0; j < 4; j++) { QATTN3_SM(false, 0, 1, j, mc2, 0); QATTN3_SM(false, 1, 1, j, mc2, 0); }
        ls0 = BYTE ? 0.0f : (acc0[0] + acc0[1]) + (acc0[2] + acc0[3]);
        ls1 = BYTE ? 0.0f : (acc1[0] + acc1[1]) + (acc1[2] + acc1[3]);
    }
    st.l_run[0] += ls0;
    st.l_run[1] += ls1;
    QATTN3_STAMP(4);
}

template <int FMT, bool CAUSAL, bool TOKEN, bool TWO, bool BYTE, int ABL = 0>
__global__ __launch_bounds__(kV3Waves * 64, 1) void attn_fwd_kernel_v3(const AttnParams p, const int qb_lo, const int qb_n) {
    constexpr int D = kV3D, CH = 64 * D, STAGE = 2 * CH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hh = lane >> 5;

    int head, qb;
    map_block(p, blockIdx.x, qb_n, CAUSAL, head, qb);
    qb += qb_lo;
    const int b = head / p.Hq, h = head % p.Hq;
    const int hkv = h / (p.Hq / p.Hkv);
    const long kv_head = (long)b * p.Hkv + hkv;
    const int q0_wg = qb * kV3QPerWG;
    const int q0 = q0_wg + wave * kV3QPerWave;  // first query row of this wave (q-block 0); q-block 1 starts at q0+32
    const unsigned char* kg = p.k + kv_head * (long)p.nchunks * CH;
    const unsigned char* vg = p.v + kv_head * (long)p.nchunks * CH;

    // chunks the workgroup / this wave must visit (causal: up to the diagonal of the last row)
    const int n_wg = CAUSAL ? min(p.nchunks, (min(q0_wg + kV3QPerWG, p.Sq) - 1) / 64 + 1) : p.nchunks;
    const int n_w = CAUSAL ? min(n_wg, (q0 + kV3QPerWave - 1) / 64 + 1) : p.nchunks;
    const int T = n_wg + 1;  // iterations t = 0 .. n_wg : QK(t) + softmax(t) + PV(t-1); the last one is PV only
    const int frag_lane_off = (hh << 10) + (ql << 4);

    WaveState3<TWO, BYTE> st;
    // Q^T fragments straight to registers
#pragma unroll
    for (int qbk = 0; qbk < 2; qbk++) {
        const int qrow = q0 + 32 * qbk + ql;
        const bool qvalid = qrow < p.Sq;
        const unsigned char* qp = p.q + (((long)b * p.Hq + h) * p.Sq + (qvalid ? qrow : 0)) * D + hh * 32;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            v4i lo = *reinterpret_cast<const v4i*>(qp + s * 64);
            v4i hi = *reinterpret_cast<const v4i*>(qp + s * 64 + 16);
            if (!qvalid) { lo = v4i{0, 0, 0, 0}; hi = v4i{0, 0, 0, 0}; }
            st.qf[qbk][s][0] = lo[0]; st.qf[qbk][s][1] = lo[1]; st.qf[qbk][s][2] = lo[2]; st.qf[qbk][s][3] = lo[3];
            st.qf[qbk][s][4] = hi[0]; st.qf[qbk][s][5] = hi[1]; st.qf[qbk][s][6] = hi[2]; st.qf[qbk][s][7] = hi[3];
        }
        // softmax scale in the exp2 domain: c = scale_q * scale_k * sm_scale * log2(e)   (tk/attention.py:204-210)
        if (TOKEN) st.c[qbk] = p.sm_log2e * (qvalid ? p.sq[((long)b * p.Hq + h) * p.Sq + qrow] : 1.0f);
        else st.c[qbk] = p.sm_log2e * p.sq[(long)b * p.Hq + h] * p.sk[kv_head];
        st.m_run[qbk] = -1.0e30f;  // finite sentinel: the first chunk always takes the fix-up branch
        st.l_run[qbk] = 0.0f;
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) st.o[qbk][m][r] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) st.l16[qbk][r] = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            st.p[0][qbk][w] = 0; st.p[1][qbk][w] = 0;
            st.pl[0][qbk][w] = 0;
            if (TWO) st.pl[TWO ? 1 : 0][qbk][w] = 0;
        }
    }
    const float* skt = TOKEN ? p.sk + kv_head * p.Skv : nullptr;

    unsigned long long dbg_t0 = 0, dbg_r0 = 0;
    if (p.dbg & 16) { dbg_t0 = __builtin_amdgcn_s_memtime(); dbg_r0 = __builtin_amdgcn_s_memrealtime(); }

    // Ring protocol (3 slots, stage(t) = {K(t), V(t-1)} in slot t%3), one barrier EARLY: stage(t) is written during
    // iteration t-2 and published by barrier(t-1), so iteration t-1 can already read K(t)'s first fragments.
    //   iteration t:  barrier(t) -> ds_write the registers holding stage(t+2) (loaded during iteration t-1) into slot
    //   (t+2)%3 == (t-1)%3 (stage(t-1) was last read in iteration t-1) -> issue the global loads of stage(t+3) -> compute.
    StageRegs3 sr;
    auto load_for = [&](int t) {
        const int kc = min(t, p.nchunks - 1), vc = min(max(t - 1, 0), p.nchunks - 1);
        stage_load3(sr, kg + (long)kc * CH, vg + (long)vc * CH, wave, lane);
    };
    load_for(0);
    stage_write3(sr, smem, wave, lane);
    if (T > 1) { load_for(1); stage_write3(sr, smem + STAGE, wave, lane); }
    if (T > 2) load_for(2);
    auto sync_iter = [&](int t) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's ds_writes are in LDS before it signals
        if (!(ABL & 2)) __builtin_amdgcn_s_barrier();
    };
    auto stage_step = [&](int t, int phase) {
        if (ABL & 1) return;
        if (phase == 0) { if (t + 2 < T) stage_write3(sr, smem + ((t + 2) % kV3Stages) * STAGE, wave, lane); }
        else { if (t + 3 < T) load_for(t + 3); }
    };
    auto full = [&](auto par_tag, int t) {
        constexpr int PAR = decltype(par_tag)::value;
        sync_iter(t);
        const unsigned char* kbuf = smem + (t % kV3Stages) * STAGE + frag_lane_off;
        const unsigned char* knext = smem + ((t + 1) % kV3Stages) * STAGE + frag_lane_off;
        auto prep = [&](int qbk, int tile) {
            prep_tile3<CAUSAL, TOKEN>(st.s[qbk][tile], tile, p, t * 64, q0 + 32 * qbk, q0 + 32 * qbk + ql, hh, skt);
        };
        QATTN3_STAMP(0);
        auto stage = [&](int phase) { stage_step(t, phase); };
        full_step3<FMT, PAR, TWO, BYTE, ABL>(st, kbuf, kbuf + CH, knext, prep, stage);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    for (int i = 0; i < 6; i++) st.seg[i] = 0;
    st.tlast = __builtin_amdgcn_s_memtime();
    int t = 0;
    {   // first K fragments (stage(0) is published by the first barrier, so peel it: barrier, then read, then step)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        st.kpre[0] = lds_read_frag(smem + frag_lane_off + (0 << 11));
        st.kpre[1] = lds_read_frag(smem + frag_lane_off + (1 << 11));
    }
    for (; t + 1 < n_w; t += 2) {
        full(P0{}, t);
        full(P1{}, t + 1);
    }
    if (t < n_w) {  // n_w odd
        full(P0{}, t);
        ++t;
    }
    // t = n_w: the last chunk's PV (stage(t) holds V(t-1))
    {
        sync_iter(t);
        stage_step(t, 0);
        stage_step(t, 1);
        const unsigned char* vbuf = smem + (t % kV3Stages) * STAGE + CH + frag_lane_off;
        const v8i v0 = lds_read_frag(vbuf + (0 << 11)), v1 = lds_read_frag(vbuf + (1 << 11));
        const v8i v2 = lds_read_frag(vbuf + (2 << 11)), v3 = lds_read_frag(vbuf + (3 << 11));
        const int par = (t - 1) & 1;  // P(t-1)
#pragma unroll
        for (int qbk = 0; qbk < 2; qbk++) {
            const v8i pp = par ? st.p[1][qbk] : st.p[0][qbk];
            st.o[qbk][0] = mfma_f8<FMT, QATTN_FMT_E4M3>(v0, pp, st.o[qbk][0]);
            st.o[qbk][1] = mfma_f8<FMT, QATTN_FMT_E4M3>(v1, pp, st.o[qbk][1]);
            st.o[qbk][2] = mfma_f8<FMT, QATTN_FMT_E4M3>(v2, pp, st.o[qbk][2]);
            st.o[qbk][3] = mfma_f8<FMT, QATTN_FMT_E4M3>(v3, pp, st.o[qbk][3]);
            if (TWO) {
                const v8i ppl = par ? st.pl[TWO ? 1 : 0][qbk] : st.pl[0][qbk];
                st.o[qbk][0] = mfma_f8<FMT, QATTN_FMT_E4M3>(v0, ppl, st.o[qbk][0]);
                st.o[qbk][1] = mfma_f8<FMT, QATTN_FMT_E4M3>(v1, ppl, st.o[qbk][1]);
                st.o[qbk][2] = mfma_f8<FMT, QATTN_FMT_E4M3>(v2, ppl, st.o[qbk][2]);
                st.o[qbk][3] = mfma_f8<FMT, QATTN_FMT_E4M3>(v3, ppl, st.o[qbk][3]);
            }
            if (BYTE) {
                v8i ones;
#pragma unroll
                for (int w = 0; w < 8; w++) ones[w] = FMT == QATTN_FMT_E4M3 ? 0x38383838 : 0x3c3c3c3c;
                st.l16[qbk] = mfma_f8<FMT, QATTN_FMT_E4M3>(ones, pp, st.l16[qbk]);
            }
        }
        ++t;
    }
    // causal: waves whose rows end earlier keep the workgroup's barrier / staging cadence until the last wave is done
    for (; t < T; ++t) { sync_iter(t); stage_step(t, 0); stage_step(t, 1); }

    if (p.dbg & 16) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            const long wid = (long)blockIdx.x * kV3Waves + wave;
            p.dbg_buf[2 * wid] = t1 - dbg_t0;
            p.dbg_buf[2 * wid + 1] = r1 - dbg_r0;
            if ((ABL & 16) && wid < 64) {
                unsigned long long* segout = p.dbg_buf + 2 * (1 << 19) + wid * 8;
                for (int i = 0; i < 6; i++) segout[i] = st.seg[i];
            }
        }
    }

    // ---- epilogue: normalise, convert, store (two q-blocks)
    const float sv = p.sv ? p.sv[kv_head] : 1.0f;
#pragma unroll
    for (int qbk = 0; qbk < 2; qbk++) {
        float l_tot;
        if (BYTE) {
            l_tot = st.l16[qbk][0];
        } else {
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(st.l_run[qbk]), __float_as_uint(st.l_run[qbk]), false, false);
            l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        const float inv = sv / l_tot;
        const int qrow = q0 + 32 * qbk + ql;
        if (qrow < p.Sq) {
            const long row_off = (((long)b * p.Hq + h) * p.Sq + qrow) * D;
            if (p.out_fmt == QATTN_FMT_BF16) {
                __bf16* op = reinterpret_cast<__bf16*>(p.out) + row_off;
#pragma unroll
                for (int m = 0; m < 4; m++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
                        bf4 tv;
#pragma unroll
                        for (int i = 0; i < 4; i++) tv[i] = (__bf16)(st.o[qbk][m][4 * j + i] * inv);
                        *reinterpret_cast<bf4*>(op + 32 * m + 8 * j + 4 * hh) = tv;
                    }
            } else {
                _Float16* op = reinterpret_cast<_Float16*>(p.out) + row_off;
#pragma unroll
                for (int m = 0; m < 4; m++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                        h4 tv;
#pragma unroll
                        for (int i = 0; i < 4; i++) tv[i] = (_Float16)(st.o[qbk][m][4 * j + i] * inv);
                        *reinterpret_cast<h4*>(op + 32 * m + 8 * j + 4 * hh) = tv;
                    }
            }
            if (p.lse && hh == 0)
                p.lse[((long)b * p.Hq + h) * p.Sq + qrow] =
                    0.6931471805599453f * (st.m_run[qbk] * st.c[qbk] - (BYTE ? kPShiftByte : kPShift)) + __logf(l_tot);
        }
    }
}

template <int FMT, bool CAUSAL, bool TOKEN, bool TWO, bool BYTE>
static int launch_v3_one(const AttnParams& p, int qb_lo, int qb_n, hipStream_t st) {
    if (qb_n <= 0) return QATTN_OK;
    const int grid = p.B * p.Hq * qb_n;
    size_t lds = (size_t)kV3Stages * 2 * 64 * kV3D;
    if (p.lds_pad > 0) lds = (size_t)p.lds_pad;
    auto kern = attn_fwd_kernel_v3<FMT, CAUSAL, TOKEN, TWO, BYTE>;
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return QATTN_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kV3Waves * 64), lds, st, p, qb_lo, qb_n);
    return QATTN_OK;
}

template <int FMT, bool CAUSAL>
static int launch_v3_t(const AttnParams& p, int scale_mode, hipStream_t st) {
    int n_two;  // leading q-blocks (of 256 rows) whose first row sees fewer than kTwoTermKeys keys -> two-term P
    if (CAUSAL) n_two = min(p.nqb, ceil_div(min(kTwoTermKeys, p.Skv), kV3QPerWG));
    else n_two = p.Skv < kTwoTermKeys ? p.nqb : 0;
    const bool byte_exp = !p.exact_exp && p.lse == nullptr;
    if (p.dbg >= 256 && !CAUSAL && scale_mode == QATTN_SCALE_HEAD && FMT == QATTN_FMT_E4M3) {
        // development: compile-time ablations of the headline kernel (QATTN_V2_DBG = 256 + mask [+16 for the cycle stamp])
        const int grid = p.B * p.Hq * p.nqb;
        const size_t lds = (size_t)kV3Stages * 2 * 64 * kV3D;
#define QATTN3_ABL_CASE(M)                                                                                      \
        case M: {                                                                                               \
            auto kern = attn_fwd_kernel_v3<QATTN_FMT_E4M3, false, false, false, true, M>;                        \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(kV3Waves * 64), lds, st, p, 0, p.nqb);                     \
            return QATTN_OK;                                                                                    \
        }
        switch ((p.dbg & 32) ? 16 : (p.dbg & 15)) {
            QATTN3_ABL_CASE(16) QATTN3_ABL_CASE(1) QATTN3_ABL_CASE(2) QATTN3_ABL_CASE(4) QATTN3_ABL_CASE(8) QATTN3_ABL_CASE(12) QATTN3_ABL_CASE(15) QATTN3_ABL_CASE(3) QATTN3_ABL_CASE(7)
            default: break;
        }
#undef QATTN3_ABL_CASE
    }
    int rc;
    if (scale_mode == QATTN_SCALE_TOKEN) {
        if (byte_exp) rc = launch_v3_one<FMT, CAUSAL, true, false, true>(p, n_two, p.nqb - n_two, st);
        else rc = launch_v3_one<FMT, CAUSAL, true, false, false>(p, n_two, p.nqb - n_two, st);
        if (rc == QATTN_OK) rc = launch_v3_one<FMT, CAUSAL, true, true, false>(p, 0, n_two, st);
    } else {
        if (byte_exp) rc = launch_v3_one<FMT, CAUSAL, false, false, true>(p, n_two, p.nqb - n_two, st);
        else rc = launch_v3_one<FMT, CAUSAL, false, false, false>(p, n_two, p.nqb - n_two, st);
        if (rc == QATTN_OK) rc = launch_v3_one<FMT, CAUSAL, false, true, false>(p, 0, n_two, st);
    }
    return rc;
}

int launch_attn_v3(const AttnParams& p, int D, int fmt, int causal, int scale_mode, hipStream_t st) {
    if (D != 128) return QATTN_ERR_UNSUPPORTED_DIM;
    if (fmt == QATTN_FMT_E4M3) return causal ? launch_v3_t<QATTN_FMT_E4M3, true>(p, scale_mode, st) : launch_v3_t<QATTN_FMT_E4M3, false>(p, scale_mode, st);
    return causal ? launch_v3_t<QATTN_FMT_E5M2, true>(p, scale_mode, st) : launch_v3_t<QATTN_FMT_E5M2, false>(p, scale_mode, st);
}

}  // namespace qattn
