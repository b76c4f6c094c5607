// qattn_attn_v3.hip -- FP8 fused attention forward for gfx950 (MI355X / CDNA4): 4 waves x 64 query rows.
//
// Replaces fwd_attend_ker<D,causal,..> + its launcher (src/quantum_attn/tk/attention.py:97-349, 355-647) behind the
// op quantum_attn::fp8_attention_forward (src/quantum_attn/ops.py:98-121).  Same algorithm and numerics as
// qattn_attn_v2.hip (swapped QK^T on v_mfma_f32_32x32x64_f8f6f4, three-deep software pipeline QK(t) / softmax(t-1) /
// PV(t-2), optimistic exponentiation with a rare fix-up, byte-exponential fast path, fragment-layout K/V), but laid
// out for what the v2 ablation showed to be the wall (profiles/r01_ablation.md): LDS-array pressure.
//
//  * workgroup = 4 waves, ONE per SIMD (up to 512 VGPRs each), each wave owns 64 query rows = two 32-row q-blocks.
//    Every K / V fragment read from LDS feeds TWO MFMAs (one per q-block): operand reads drop from
//    8 waves x 22 to 4 waves x 16 ds_read_b128 per 64-key chunk (704 -> 256 LDS-array cycles).
//  * Q^T fragments (2 q-blocks x 2 k-steps) live in registers for the whole sweep.
//  * 18 hand-placed MFMA slots per iteration: 8 PV (+2 row-sum) + 8 QK^T, each with a 4-score softmax slice
//    underneath; fragments are requested two or more slots ahead; the first two V fragments of the next
//    iteration are read during the current one (their stage has been visible since the last barrier).
//  * K/V staging through registers (global_load_dwordx4 one iteration ahead, ds_write_b128 after the next barrier),
//    3-stage ring, stage(t) = {K chunk t, V chunk t-1}, one s_barrier per iteration among 4 waves.
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
    v16f s[2][2][2];         // S^T ping-pong [t&1][q-block][32-key tile]
    v8i p[2][2];             // P^T (e4m3) ping-pong [t&1][q-block]
    v8i pl[TWO ? 2 : 1][2];  // low term of the two-term split
    v8i qf[2][2];            // Q^T fragments [q-block][k-step]
    v8i vpre[2];             // V fragments (row blocks 0,1) of the NEXT iteration's PV
    v16f l16[2];             // BYTE: row sums of the quantised P' (every register holds the full sum)
    float m_run[2];          // running max of the raw scores
    float l_run[2];          // exact mode: this lane's partial row sums
    float c[2];              // scale_q*scale_k*sm_scale*log2(e) per q-block row
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

// token-wise key scales and the ragged-tail / causal-diagonal mask, in place on a finished S^T chunk of q-block b
template <bool CAUSAL, bool TOKEN>
__device__ __forceinline__ void prep_scores3(v16f& s0, v16f& s1, const AttnParams& p, int k0, int q0b, int qrow, int hh, const float* skt) {
    if (TOKEN) {
#pragma unroll
        for (int tt = 0; tt < 2; tt++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int kk = k0 + 32 * tt + 8 * j + 4 * hh;
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kk + 3 < p.Skv) w = *reinterpret_cast<const float4*>(skt + kk);
                else { if (kk < p.Skv) w.x = skt[kk]; if (kk + 1 < p.Skv) w.y = skt[kk + 1]; if (kk + 2 < p.Skv) w.z = skt[kk + 2]; }
                v16f& sx = tt ? s1 : s0;
                sx[4 * j + 0] *= w.x; sx[4 * j + 1] *= w.y; sx[4 * j + 2] *= w.z; sx[4 * j + 3] *= w.w;
            }
    }
    const bool need_mask = (k0 + 64 > p.Skv) || (CAUSAL && k0 + 63 > q0b);  // wave-uniform
    if (__builtin_expect(need_mask, 0)) {
#pragma unroll
        for (int r = 0; r < 32; r++) {
            const int key = k0 + 32 * (r >> 4) + (r & 3) + 8 * ((r & 15) >> 2) + 4 * hh;
            const bool dead = key >= p.Skv || (CAUSAL && key > qrow);
            v16f& sx = (r >> 4) ? s1 : s0;
            sx[r & 15] = dead ? -INFINITY : sx[r & 15];
        }
    }
}

#define QATTN3_FENCE() __builtin_amdgcn_sched_barrier(0)
// softmax group: q-block B, tile TL, group J of 4 scores -> P dword 4*TL+J
#define QATTN3_SM(FIRST, B, TL, J, MC, SEED)                                                             \
    do {                                                                                                 \
        if (BYTE) byte_group3(TL ? sc##B##1 : sc##B##0, J, cx[B], MC[B], pc##B, 4 * (TL) + (J), SEED);    \
        else exp_group3<TWO, FIRST>(TL ? sc##B##1 : sc##B##0, J, cx[B], MC[B], acc##B, pc##B, pcl##B, 4 * (TL) + (J), SEED); \
    } while (0)
#define QATTN3_PV(M, FR)                                                                                 \
    do {                                                                                                 \
        st.o[0][M] = mfma_f8<FMT, QATTN_FMT_E4M3>(FR, pp0, st.o[0][M]);                                   \
        if (TWO) st.o[0][M] = mfma_f8<FMT, QATTN_FMT_E4M3>(FR, ppl0, st.o[0][M]);                         \
    } while (0)
#define QATTN3_PV1(M, FR)                                                                                \
    do {                                                                                                 \
        st.o[1][M] = mfma_f8<FMT, QATTN_FMT_E4M3>(FR, pp1, st.o[1][M]);                                   \
        if (TWO) st.o[1][M] = mfma_f8<FMT, QATTN_FMT_E4M3>(FR, ppl1, st.o[1][M]);                         \
    } while (0)

// One pipelined iteration (1 <= t <= n_w): PV(t-2), [row sums], QK(t), softmax(t-1).  PAR = t & 1.
//   kbuf  : stage(t) K part (+ lane offset)     vprev : stage(t-1) V part = V(t-2)     vnext : stage(t) V part = V(t-1)
template <int FMT, int PAR, bool TWO, bool BYTE>
__device__ __forceinline__ void full_step3(WaveState3<TWO, BYTE>& st, const unsigned char* kbuf, const unsigned char* vprev,
                                           const unsigned char* vnext) {
    constexpr int PL_R = TWO ? PAR : 0, PL_W = TWO ? (PAR ^ 1) : 0;
    v16f& sn00 = st.s[PAR][0][0]; v16f& sn01 = st.s[PAR][0][1];  // S(t)   q-block 0 tiles
    v16f& sn10 = st.s[PAR][1][0]; v16f& sn11 = st.s[PAR][1][1];  //        q-block 1 tiles
    const v16f& sc00 = st.s[PAR ^ 1][0][0]; const v16f& sc01 = st.s[PAR ^ 1][0][1];  // S(t-1)
    const v16f& sc10 = st.s[PAR ^ 1][1][0]; const v16f& sc11 = st.s[PAR ^ 1][1][1];
    v8i& pc0 = st.p[PAR ^ 1][0]; v8i& pc1 = st.p[PAR ^ 1][1];          // P(t-1) being produced
    v8i& pcl0 = st.pl[PL_W][0]; v8i& pcl1 = st.pl[PL_W][1];
    const v8i& pp0 = st.p[PAR][0]; const v8i& pp1 = st.p[PAR][1];      // P(t-2) consumed by PV
    const v8i& ppl0 = st.pl[PL_R][0]; const v8i& ppl1 = st.pl[PL_R][1];
    constexpr float SHIFT = BYTE ? kPShiftByte : kPShift, THR = BYTE ? kRescaleThrByte : kRescaleThr;
    float cx[2], mc[2];
#pragma unroll
    for (int b = 0; b < 2; b++) {
        cx[b] = BYTE ? 8.0f * st.c[b] : st.c[b];
        mc[b] = BYTE ? __builtin_fmaf(-8.0f * st.m_run[b], st.c[b], 8.0f * SHIFT + 56.0f + kByteBias) : SHIFT - st.m_run[b] * st.c[b];
    }
    float acc0[4], acc1[4];

    // ---- PV(t-2): 8 slots; V0,V1 were read last iteration
    QATTN3_PV(0, st.vpre[0]);                                   // slot 0
    v8i f2 = lds_read_frag(vprev + (2 << 11));
    float mx0 = tile_max(sc00, sc01);
    QATTN3_FENCE();
    QATTN3_PV1(0, st.vpre[0]);                                  // slot 1
    float mx1 = tile_max(sc10, sc11);
    QATTN3_FENCE();
    QATTN3_PV(1, st.vpre[1]);                                   // slot 2
    v8i f3 = lds_read_frag(vprev + (3 << 11));
    QATTN3_SM(true, 0, 0, 0, mc, pp0[0]);
    QATTN3_FENCE();
    QATTN3_PV1(1, st.vpre[1]);                                  // slot 3
    QATTN3_SM(true, 1, 0, 0, mc, pp1[0]);
    QATTN3_FENCE();
    QATTN3_PV(2, f2);                                           // slot 4
    v8i ka = lds_read_frag(kbuf + (0 << 11));                   // K (tile 0, k-step 0)
    QATTN3_SM(false, 0, 0, 1, mc, pc0[0]);
    QATTN3_FENCE();
    QATTN3_PV1(2, f2);                                          // slot 5
    QATTN3_SM(false, 1, 0, 1, mc, pc1[0]);
    QATTN3_FENCE();
    QATTN3_PV(3, f3);                                           // slot 6
    v8i kb = lds_read_frag(kbuf + (2 << 11));                   // K (tile 1, k-step 0)
    QATTN3_SM(false, 0, 0, 2, mc, pc0[1]);
    QATTN3_FENCE();
    QATTN3_PV1(3, f3);                                          // slot 7
    QATTN3_SM(false, 1, 0, 2, mc, pc1[1]);
    QATTN3_FENCE();
    // ---- BYTE: row sums of the quantised P(t-2) on the matrix pipe (ones(32x64) . P^T)
    if (BYTE) {
        v8i ones;
#pragma unroll
        for (int w = 0; w < 8; w++) ones[w] = FMT == QATTN_FMT_E4M3 ? 0x38383838 : 0x3c3c3c3c;  // 1.0 in e4m3 / e5m2
        st.l16[0] = mfma_f8<FMT, QATTN_FMT_E4M3>(ones, pp0, st.l16[0]);   // slot 8
        QATTN3_SM(false, 0, 0, 3, mc, pc0[2]);
        QATTN3_FENCE();
        st.l16[1] = mfma_f8<FMT, QATTN_FMT_E4M3>(ones, pp1, st.l16[1]);   // slot 9
        QATTN3_SM(false, 1, 0, 3, mc, pc1[2]);
        QATTN3_FENCE();
    } else {
        QATTN3_SM(false, 0, 0, 3, mc, pc0[2]);
        QATTN3_SM(false, 1, 0, 3, mc, pc1[2]);
    }
    // ---- QK^T(t): 8 slots
#pragma unroll
    for (int r = 0; r < 16; r++) { sn00[r] = 0.0f; sn01[r] = 0.0f; sn10[r] = 0.0f; sn11[r] = 0.0f; }
    sn00 = mfma_f8<FMT, FMT>(ka, st.qf[0][0], sn00);            // slot 10: S0[q0] = K(0,0).Q0
    v8i kc = lds_read_frag(kbuf + (1 << 11));                   // K (tile 0, k-step 1)
    QATTN3_SM(false, 0, 1, 0, mc, pc0[3]);
    QATTN3_FENCE();
    sn10 = mfma_f8<FMT, FMT>(ka, st.qf[1][0], sn10);            // slot 11: S0[q1]
    QATTN3_SM(false, 1, 1, 0, mc, pc1[3]);
    QATTN3_FENCE();
    sn01 = mfma_f8<FMT, FMT>(kb, st.qf[0][0], sn01);            // slot 12: S1[q0] = K(1,0).Q0
    v8i kd = lds_read_frag(kbuf + (3 << 11));                   // K (tile 1, k-step 1)
    QATTN3_SM(false, 0, 1, 1, mc, pc0[4]);
    QATTN3_FENCE();
    sn11 = mfma_f8<FMT, FMT>(kb, st.qf[1][0], sn11);            // slot 13
    QATTN3_SM(false, 1, 1, 1, mc, pc1[4]);
    QATTN3_FENCE();
    sn00 = mfma_f8<FMT, FMT>(kc, st.qf[0][1], sn00);            // slot 14: S0[q0] += K(0,1).Q1
    st.vpre[0] = lds_read_frag(vnext + (0 << 11));              // next iteration's V0
    QATTN3_SM(false, 0, 1, 2, mc, pc0[5]);
    QATTN3_FENCE();
    sn10 = mfma_f8<FMT, FMT>(kc, st.qf[1][1], sn10);            // slot 15
    QATTN3_SM(false, 1, 1, 2, mc, pc1[5]);
    QATTN3_FENCE();
    sn01 = mfma_f8<FMT, FMT>(kd, st.qf[0][1], sn01);            // slot 16
    st.vpre[1] = lds_read_frag(vnext + (1 << 11));              // next iteration's V1
    QATTN3_SM(false, 0, 1, 3, mc, pc0[6]);
    QATTN3_FENCE();
    sn11 = mfma_f8<FMT, FMT>(kd, st.qf[1][1], sn11);            // slot 17
    QATTN3_SM(false, 1, 1, 3, mc, pc1[6]);
    {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx0), __float_as_uint(mx0), false, false);
        mx0 = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx1), __float_as_uint(mx1), false, false);
        mx1 = fmaxf(__uint_as_float(sx[0]), __uint_as_float(sx[1]));
    }
    float ls0 = BYTE ? 0.0f : (acc0[0] + acc0[1]) + (acc0[2] + acc0[3]);
    float ls1 = BYTE ? 0.0f : (acc1[0] + acc1[1]) + (acc1[2] + acc1[3]);
    QATTN3_FENCE();
    // ---- rare fix-up (always on the first chunk): rescale what is accumulated, redo this chunk's exponentials
    const bool need0 = (mx0 - st.m_run[0]) * st.c[0] > THR, need1 = (mx1 - st.m_run[1]) * st.c[1] > THR;
    if (__builtin_expect(__any(need0 || need1) != 0, 0)) {
        float mc2[2];
        {
            const float m_new = fmaxf(st.m_run[0], mx0);
            const float alpha = __builtin_amdgcn_exp2f((st.m_run[0] - m_new) * st.c[0]);
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int r = 0; r < 16; r++) st.o[0][m][r] *= alpha;
            st.l_run[0] *= alpha;
            if (BYTE) {
#pragma unroll
                for (int r = 0; r < 16; r++) st.l16[0][r] *= alpha;
            }
            st.m_run[0] = m_new;
            mc2[0] = BYTE ? __builtin_fmaf(-8.0f * m_new, st.c[0], 8.0f * SHIFT + 56.0f + kByteBias) : SHIFT - m_new * st.c[0];
        }
        {
            const float m_new = fmaxf(st.m_run[1], mx1);
            const float alpha = __builtin_amdgcn_exp2f((st.m_run[1] - m_new) * st.c[1]);
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int r = 0; r < 16; r++) st.o[1][m][r] *= alpha;
            st.l_run[1] *= alpha;
            if (BYTE) {
#pragma unroll
                for (int r = 0; r < 16; r++) st.l16[1][r] *= alpha;
            }
            st.m_run[1] = m_new;
            mc2[1] = BYTE ? __builtin_fmaf(-8.0f * m_new, st.c[1], 8.0f * SHIFT + 56.0f + kByteBias) : SHIFT - m_new * st.c[1];
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
}

template <int FMT, bool CAUSAL, bool TOKEN, bool TWO, bool BYTE>
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
    const int T = n_wg + 2;  // iterations t = 0 .. n_wg+1 : QK(t), softmax(t-1), PV(t-2)
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

    // Ring protocol (3 stages, stage(t) = {K(t), V(t-1)} in slot t%3): iteration t = barrier(t) -> ds_write the
    // registers holding stage(t+1) (loaded during iteration t-1) -> issue the global loads of stage(t+2) -> compute.
    StageRegs3 sr;
    auto load_for = [&](int t) {
        const int kc = min(t, p.nchunks - 1), vc = min(max(t - 1, 0), p.nchunks - 1);
        stage_load3(sr, kg + (long)kc * CH, vg + (long)vc * CH, wave, lane);
    };
    load_for(0);
    stage_write3(sr, smem, wave, lane);
    load_for(1);
    auto sync_iter = [&](int t) -> const unsigned char* {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's ds_writes of stage(t) are in LDS
        __builtin_amdgcn_s_barrier();
        if (t + 1 < T) stage_write3(sr, smem + ((t + 1) % kV3Stages) * STAGE, wave, lane);
        if (t + 2 < T) load_for(t + 2);
        return smem + (t % kV3Stages) * STAGE + frag_lane_off;
    };
    auto full = [&](auto par_tag, int t) {
        constexpr int PAR = decltype(par_tag)::value;
        const unsigned char* kbuf = sync_iter(t);
        const unsigned char* vprev = smem + ((t - 1) % kV3Stages) * STAGE + CH + frag_lane_off;
        prep_scores3<CAUSAL, TOKEN>(st.s[PAR ^ 1][0][0], st.s[PAR ^ 1][0][1], p, (t - 1) * 64, q0, q0 + ql, hh, skt);
        prep_scores3<CAUSAL, TOKEN>(st.s[PAR ^ 1][1][0], st.s[PAR ^ 1][1][1], p, (t - 1) * 64, q0 + 32, q0 + 32 + ql, hh, skt);
        full_step3<FMT, PAR, TWO, BYTE>(st, kbuf, vprev, kbuf + CH);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    // t = 0: QK(0) only (+ the first V prefetch; it is multiplied by P = 0 at t = 1)
    {
        const unsigned char* kbuf = sync_iter(0);
#pragma unroll
        for (int r = 0; r < 16; r++) { st.s[0][0][0][r] = 0.f; st.s[0][0][1][r] = 0.f; st.s[0][1][0][r] = 0.f; st.s[0][1][1][r] = 0.f; }
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const v8i ka = lds_read_frag(kbuf + ((0 * 2 + s) << 11));
            const v8i kb = lds_read_frag(kbuf + ((1 * 2 + s) << 11));
            st.s[0][0][0] = mfma_f8<FMT, FMT>(ka, st.qf[0][s], st.s[0][0][0]);
            st.s[0][1][0] = mfma_f8<FMT, FMT>(ka, st.qf[1][s], st.s[0][1][0]);
            st.s[0][0][1] = mfma_f8<FMT, FMT>(kb, st.qf[0][s], st.s[0][0][1]);
            st.s[0][1][1] = mfma_f8<FMT, FMT>(kb, st.qf[1][s], st.s[0][1][1]);
        }
        st.vpre[0] = lds_read_frag(kbuf + CH + (0 << 11));
        st.vpre[1] = lds_read_frag(kbuf + CH + (1 << 11));
    }
    int t = 1;
    for (; t + 1 <= n_w; t += 2) {
        full(P1{}, t);
        full(P0{}, t + 1);
    }
    if (t <= n_w) {  // n_w odd
        full(P1{}, t);
        ++t;
    }
    // t = n_w + 1: the last chunk's PV (V(t-2) lives in stage(t-1); row blocks 0,1 are already in vpre)
    {
        (void)sync_iter(t);
        const unsigned char* vprev = smem + ((t - 1) % kV3Stages) * STAGE + CH + frag_lane_off;
        const v8i f2 = lds_read_frag(vprev + (2 << 11)), f3 = lds_read_frag(vprev + (3 << 11));
        const int par = t & 1;
#pragma unroll
        for (int qbk = 0; qbk < 2; qbk++) {
            const v8i pp = par ? st.p[1][qbk] : st.p[0][qbk];
            st.o[qbk][0] = mfma_f8<FMT, QATTN_FMT_E4M3>(st.vpre[0], pp, st.o[qbk][0]);
            st.o[qbk][1] = mfma_f8<FMT, QATTN_FMT_E4M3>(st.vpre[1], pp, st.o[qbk][1]);
            st.o[qbk][2] = mfma_f8<FMT, QATTN_FMT_E4M3>(f2, pp, st.o[qbk][2]);
            st.o[qbk][3] = mfma_f8<FMT, QATTN_FMT_E4M3>(f3, pp, st.o[qbk][3]);
            if (TWO) {
                const v8i ppl = par ? st.pl[TWO ? 1 : 0][qbk] : st.pl[0][qbk];
                st.o[qbk][0] = mfma_f8<FMT, QATTN_FMT_E4M3>(st.vpre[0], ppl, st.o[qbk][0]);
                st.o[qbk][1] = mfma_f8<FMT, QATTN_FMT_E4M3>(st.vpre[1], ppl, st.o[qbk][1]);
                st.o[qbk][2] = mfma_f8<FMT, QATTN_FMT_E4M3>(f2, ppl, st.o[qbk][2]);
                st.o[qbk][3] = mfma_f8<FMT, QATTN_FMT_E4M3>(f3, ppl, st.o[qbk][3]);
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
    for (; t < T; ++t) sync_iter(t);

    if (p.dbg & 16) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            const long wid = (long)blockIdx.x * kV3Waves + wave;
            p.dbg_buf[2 * wid] = t1 - dbg_t0;
            p.dbg_buf[2 * wid + 1] = r1 - dbg_r0;
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
