// qattn_attn_v2.hip -- FP8 fused attention forward for gfx950 (MI355X / CDNA4), software-pipelined structure.
//
// Replaces fwd_attend_ker<D,causal,..> + its launcher (src/quantum_attn/tk/attention.py:97-349, 355-647) behind the
// op quantum_attn::fp8_attention_forward (src/quantum_attn/ops.py:98-121).  Designed for CDNA4, not translated:
//
//  * workgroup = 8 waves (2 per SIMD) = 256 query rows; each wave owns 32 query rows for the whole KV sweep.
//  * both GEMMs on v_mfma_f32_32x32x64_f8f6f4 (unscaled form = full FP8 rate, profiles/r01_mfma_probe.log):
//        S^T[key][q] = K . Q^T      (A = K fragment from LDS, B = Q^T fragment held in registers)
//        O^T[d][q]  += V^T . P^T    (A = V^T fragment from LDS, B = P^T built in registers from S^T)
//    The swapped orientation puts the query on the LANE (col = lane&31) and the keys in the accumulator registers
//    (row = (r&3) + 8*(r>>2) + 4*(lane>>5)): softmax statistics are per-lane scalars, the only cross-lane traffic
//    is one v_permlane32_swap per chunk, and the fp8-converted P registers ARE the next MFMA's B operand.
//  * K and V arrive pre-laid in fragment order (include/qattn.h): a 64-key chunk is a linear LDS-DMA copy
//    (global_load_lds_dwordx4) and every operand read is a conflict-free ds_read_b128.
//  * three-deep software pipeline inside every wave: iteration t issues the QK^T MFMAs of chunk t and the PV MFMAs
//    of chunk t-2 while the VALU runs the softmax of chunk t-1 -- one basic block of 8 MFMAs + ~110 VALU ops, so the
//    matrix pipe (64 cycles per MFMA) and the softmax (v_exp_f32 / v_cvt_pk_fp8_f32 are ~9-cycle issues,
//    profiles/r01_mfma_valu_probe.log) overlap inside one wave and across the two waves of a SIMD.
//  * the softmax exponentiates OPTIMISTICALLY against the running max m_run; only if some row's chunk max exceeds
//    m_run by more than kRescaleThr (so P' could overflow e4m3) a rare fix-up branch rescales O and l and redoes the
//    chunk's exponentials.  The common path has no branch between the MFMAs and the VALU work.
//  * LDS ring of 3 stages, stage(t) = {K chunk t, V chunk t-2}, one s_barrier per iteration, DMA two iterations
//    ahead behind a counted vmcnt.
//  * P is scaled by 2^kPShift before the e4m3 conversion; where few keys are visible P is split hi+lo (two terms).
#include <type_traits>

#include "qattn_attn.h"

namespace qattn {

constexpr int kStagesV2 = 3;

template <int D>
__device__ __forceinline__ void stage_kv(const unsigned char* ksrc, const unsigned char* vsrc, unsigned char* lds_stage,
                                         int wave, int lane) {
    // [K chunk | V chunk] = 2*64*D bytes, linear; every wave-instruction moves 1 KiB (64 lanes x 16 B)
    constexpr int CH = 64 * D;
    constexpr int ROUNDS = 2 * CH / (kThreads * 16);
    const int wave_base = wave << 10;  // wave-uniform
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const int o = r * (kThreads * 16) + wave_base;
        const unsigned char* src = (o < CH ? ksrc + o : vsrc + (o - CH)) + (lane << 4);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds_stage + o), 16, 0, 0);
    }
}

template <int D, bool TWO>
struct WaveState {
    static constexpr int MB = D / 32;
    v16f o[MB];             // O^T accumulators
    v16f s[2][2];           // S^T ping-pong: s[t&1] holds chunk t's two 32-key tiles
    v8i p[2];               // P^T (e4m3) ping-pong: p[t&1] holds chunk t
    v8i pl[TWO ? 2 : 1];    // low term of the two-term split (unused when !TWO)
    float m_run;   // running max of the raw scores
    float l_run;   // this lane's partial row sum of P'
    float c;       // scale_q*scale_k*sm_scale*log2(e)
};

template <int QK_FMT, int D>
__device__ __forceinline__ void qk_chunk(const unsigned char* kbuf, const unsigned char* qbuf, v16f& s0, v16f& s1) {
    constexpr int KS = D / 64;
#pragma unroll
    for (int r = 0; r < 16; r++) { s0[r] = 0.0f; s1[r] = 0.0f; }
#pragma unroll
    for (int s = 0; s < KS; s++) {
        const v8i qf = lds_read_frag(qbuf + (s << 11));
        const v8i ka = lds_read_frag(kbuf + ((0 * KS + s) << 11));
        const v8i kb = lds_read_frag(kbuf + ((1 * KS + s) << 11));
        s0 = mfma_f8<QK_FMT, QK_FMT>(ka, qf, s0);
        s1 = mfma_f8<QK_FMT, QK_FMT>(kb, qf, s1);
    }
}

template <int V_FMT, int D, bool TWO>
__device__ __forceinline__ void pv_chunk(const unsigned char* vbuf, const v8i& ph, const v8i& plo, v16f (&o)[D / 32]) {
#pragma unroll
    for (int m = 0; m < D / 32; m++) {
        const v8i va = lds_read_frag(vbuf + (m << 11));
        o[m] = mfma_f8<V_FMT, QATTN_FMT_E4M3>(va, ph, o[m]);
        if (TWO) o[m] = mfma_f8<V_FMT, QATTN_FMT_E4M3>(va, plo, o[m]);
    }
}

// In-place fix-ups of a finished S^T chunk before its softmax (both rare or cheap, kept out of the hot block):
// token-wise key scales (inductor/kernels/attention.py:395) and the ragged-tail / causal-diagonal mask.
template <bool CAUSAL, bool TOKEN>
__device__ __forceinline__ void prep_scores(v16f& s0, v16f& s1, const AttnParams& p, int k0, int q0, int qrow, int hh,
                                            const float* skt) {
    if (TOKEN) {
#pragma unroll
        for (int tt = 0; tt < 2; tt++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int kk = k0 + 32 * tt + 8 * j + 4 * hh;  // keys kk..kk+3 live in registers 4j..4j+3 of tile tt
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kk + 3 < p.Skv) w = *reinterpret_cast<const float4*>(skt + kk);
                else { if (kk < p.Skv) w.x = skt[kk]; if (kk + 1 < p.Skv) w.y = skt[kk + 1]; if (kk + 2 < p.Skv) w.z = skt[kk + 2]; }
                v16f& sx = tt ? s1 : s0;
                sx[4 * j + 0] *= w.x; sx[4 * j + 1] *= w.y; sx[4 * j + 2] *= w.z; sx[4 * j + 3] *= w.w;
            }
    }
    const bool need_mask = (k0 + 64 > p.Skv) || (CAUSAL && k0 + 63 > q0);  // wave-uniform
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

// 4 scores -> 4 exponentials -> one dword of the e4m3 P operand (+ the residual dword when TWO); accumulates the
// partial row sums in acc[0..3] (FIRST: initialises them).  `seed` only provides the register the first
// v_cvt_pk_fp8_f32 writes its low half into (its high half is overwritten by the second), saving a v_mov.
template <bool TWO, bool FIRST>
__device__ __forceinline__ void exp_group(const v16f& sx, int j, float c, float mc, float (&acc)[4], v8i& pv, v8i& plv,
                                          int w, int seed) {
    float e[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        e[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[4 * j + i], c, mc));
        acc[i] = FIRST ? e[i] : acc[i] + e[i];
    }
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));  // sums stay in this slot
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
    asm volatile("" : "+v"(ph));  // keep the optimistic result materialised HERE (under the MFMA), not sunk past the fix-up branch
    pv[w] = ph;
}

#define QATTN_SLOT_FENCE() __builtin_amdgcn_sched_barrier(0)

// One pipelined iteration (1 <= t <= n_w): QK(t), softmax(t-1), PV(t-2), hand-placed in 8 MFMA slots.  PAR = t & 1.
// Slot i = { MFMA i ; ds_reads of the fragment(s) slot i+1 needs ; softmax group i (4 scores) } -- the MFMA is first
// in program order so the loads and the VALU slice issue underneath it.  QK^T goes first and PV last: S(t) is then
// complete long before the next iteration's VALU reads it, and only the rare fix-up waits for the PV accumulators.
// The Q^T fragments are parked in LDS (each lane re-reads its own 64 bytes per iteration): registers, not LDS
// bandwidth, are the scarce resource at two waves per SIMD.
template <int D, int QK_FMT, int V_FMT, int PAR, bool TWO>
__device__ __forceinline__ void full_step(WaveState<D, TWO>& st, const unsigned char* kbuf, const unsigned char* qbuf) {
    static_assert(D == 128, "hand-placed slots are written for D = 128");
    constexpr int CH = 64 * D;
    constexpr int PL_R = TWO ? PAR : 0, PL_W = TWO ? (PAR ^ 1) : 0;
    const unsigned char* vbuf = kbuf + CH;
    v16f& sn0 = st.s[PAR][0];            // S(t)   tile 0 (keys  0..31 of chunk t)
    v16f& sn1 = st.s[PAR][1];            //        tile 1 (keys 32..63)
    const v16f& sc0 = st.s[PAR ^ 1][0];  // S(t-1) tiles: the chunk being exponentiated
    const v16f& sc1 = st.s[PAR ^ 1][1];
    v8i& pc = st.p[PAR ^ 1];             // P(t-1) being produced
    v8i& pcl = st.pl[PL_W];
    const v8i& pp = st.p[PAR];           // P(t-2) consumed by PV
    const v8i& ppl = st.pl[PL_R];
    const float c = st.c, mc = kPShift - st.m_run * st.c;
    float acc[4];

    // pre: first operands in flight, chunk max of S(t-1) underneath their latency
    v8i qf = lds_read_frag(qbuf);              // Q^T k-step 0
    v8i fa = lds_read_frag(kbuf + (0 << 11));  // K (tile 0, k-step 0)
    float mx = fmaxf(fmaxf(sc0[0], sc0[1]), sc0[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, sc0[r]), sc0[r + 1]);
    mx = fmaxf(fmaxf(mx, sc0[15]), sc1[0]);
#pragma unroll
    for (int r = 1; r < 15; r += 2) mx = fmaxf(fmaxf(mx, sc1[r]), sc1[r + 1]);
    mx = fmaxf(mx, sc1[15]);
    QATTN_SLOT_FENCE();
    // slot 0: S0 = K(0,0).Q0
#pragma unroll
    for (int r = 0; r < 16; r++) { sn0[r] = 0.0f; sn1[r] = 0.0f; }
    sn0 = mfma_f8<QK_FMT, QK_FMT>(fa, qf, sn0);
    v8i fb = lds_read_frag(kbuf + (2 << 11));  // K (tile 1, k-step 0)
    exp_group<TWO, true>(sc0, 0, c, mc, acc, pc, pcl, 0, pp[0]);
    QATTN_SLOT_FENCE();
    // slot 1: S1 = K(1,0).Q0
    sn1 = mfma_f8<QK_FMT, QK_FMT>(fb, qf, sn1);
    qf = lds_read_frag(qbuf + (1 << 11));      // Q^T k-step 1
    fa = lds_read_frag(kbuf + (1 << 11));      // K (tile 0, k-step 1)
    exp_group<TWO, false>(sc0, 1, c, mc, acc, pc, pcl, 1, pc[0]);
    QATTN_SLOT_FENCE();
    // slot 2: S0 += K(0,1).Q1
    sn0 = mfma_f8<QK_FMT, QK_FMT>(fa, qf, sn0);
    fb = lds_read_frag(kbuf + (3 << 11));      // K (tile 1, k-step 1)
    exp_group<TWO, false>(sc0, 2, c, mc, acc, pc, pcl, 2, pc[1]);
    QATTN_SLOT_FENCE();
    // slot 3: S1 += K(1,1).Q1
    sn1 = mfma_f8<QK_FMT, QK_FMT>(fb, qf, sn1);
    fa = lds_read_frag(vbuf + (0 << 11));      // V block 0
    exp_group<TWO, false>(sc0, 3, c, mc, acc, pc, pcl, 3, pc[2]);
    QATTN_SLOT_FENCE();
    // slot 4: O0 += V0.P(t-2)
    st.o[0] = mfma_f8<V_FMT, QATTN_FMT_E4M3>(fa, pp, st.o[0]);
    if (TWO) st.o[0] = mfma_f8<V_FMT, QATTN_FMT_E4M3>(fa, ppl, st.o[0]);
    fb = lds_read_frag(vbuf + (1 << 11));
    exp_group<TWO, false>(sc1, 0, c, mc, acc, pc, pcl, 4, pc[3]);
    QATTN_SLOT_FENCE();
    // slot 5
    st.o[1] = mfma_f8<V_FMT, QATTN_FMT_E4M3>(fb, pp, st.o[1]);
    if (TWO) st.o[1] = mfma_f8<V_FMT, QATTN_FMT_E4M3>(fb, ppl, st.o[1]);
    fa = lds_read_frag(vbuf + (2 << 11));
    exp_group<TWO, false>(sc1, 1, c, mc, acc, pc, pcl, 5, pc[4]);
    QATTN_SLOT_FENCE();
    // slot 6
    st.o[2] = mfma_f8<V_FMT, QATTN_FMT_E4M3>(fa, pp, st.o[2]);
    if (TWO) st.o[2] = mfma_f8<V_FMT, QATTN_FMT_E4M3>(fa, ppl, st.o[2]);
    fb = lds_read_frag(vbuf + (3 << 11));
    exp_group<TWO, false>(sc1, 2, c, mc, acc, pc, pcl, 6, pc[5]);
    QATTN_SLOT_FENCE();
    // slot 7
    st.o[3] = mfma_f8<V_FMT, QATTN_FMT_E4M3>(fb, pp, st.o[3]);
    if (TWO) st.o[3] = mfma_f8<V_FMT, QATTN_FMT_E4M3>(fb, ppl, st.o[3]);
    exp_group<TWO, false>(sc1, 3, c, mc, acc, pc, pcl, 7, pc[6]);
    {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    float ls = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    QATTN_SLOT_FENCE();
    // rare fix-up: some row's max grew by more than the threshold (always on the first chunk: m_run = -1e30):
    // rescale everything accumulated so far (O includes PV(t-2)) and redo this chunk's exponentials
    if (__builtin_expect(__any((mx - st.m_run) * c > kRescaleThr) != 0, 0)) {
        const float m_new = fmaxf(st.m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((st.m_run - m_new) * c);
#pragma unroll
        for (int m = 0; m < D / 32; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) st.o[m][r] *= alpha;
        st.l_run *= alpha;
        st.m_run = m_new;
        const float mc2 = kPShift - m_new * c;
        exp_group<TWO, true>(sc0, 0, c, mc2, acc, pc, pcl, 0, 0);
#pragma unroll
        for (int j = 1; j < 4; j++) exp_group<TWO, false>(sc0, j, c, mc2, acc, pc, pcl, j, 0);
#pragma unroll
        for (int j = 0; j < 4; j++) exp_group<TWO, false>(sc1, j, c, mc2, acc, pc, pcl, 4 + j, 0);
        ls = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
    st.l_run += ls;
}

// The KV sweep of one wave.  Returns with st.o / st.l_run / st.m_run final.
template <int D, int QK_FMT, int V_FMT, bool CAUSAL, bool TOKEN, bool TWO>
__device__ __forceinline__ void kv_sweep(WaveState<D, TWO>& st, const AttnParams& p, unsigned char* smem,
                                         const unsigned char* kg, const unsigned char* vg, const unsigned char* qbuf, int n_wg,
                                         int n_w, int q0, int qrow, int wave, int lane, const float* skt) {
    constexpr int CH = 64 * D, STAGE = 2 * CH;
    constexpr int ROUNDS = 2 * CH / (kThreads * 16);
    const int hh = lane >> 5;
    const int T = n_wg + 2;  // iterations t = 0 .. n_wg+1 : QK(t), softmax(t-1), PV(t-2)
    const int frag_lane_off = (hh << 10) + ((lane & 31) << 4);
#pragma unroll
    for (int w = 0; w < 8; w++) {
        st.p[0][w] = 0; st.p[1][w] = 0;
        st.pl[0][w] = 0;
        if (TWO) st.pl[TWO ? 1 : 0][w] = 0;
    }
    auto stage_for = [&](int t) {
        const int kc = min(t, p.nchunks - 1), vc = min(max(t - 2, 0), p.nchunks - 1);
        stage_kv<D>(kg + (long)kc * CH, vg + (long)vc * CH, smem + (t % kStagesV2) * STAGE, wave, lane);
    };
    // every iteration t (all T of them, on every wave) starts with: stage(t) landed -> barrier -> refill stage(t+2)
    auto sync_iter = [&](int t) -> const unsigned char* {
        // stage(t)'s DMA was issued two iterations ago; at most stage(t+1)'s may stay in flight
        if (t + 1 < T) { if (ROUNDS == 1) wait_vmcnt<1>(); else if (ROUNDS == 2) wait_vmcnt<2>(); else wait_vmcnt<4>(); }
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        // every wave is past its reads of stage (t-1)%3 == (t+2)%3: refill it
        if (t + 2 < T) stage_for(t + 2);
        return smem + (t % kStagesV2) * STAGE + frag_lane_off;
    };
    auto full = [&](auto par_tag, int t) {
        constexpr int PAR = decltype(par_tag)::value;
        const unsigned char* kbuf = sync_iter(t);
        prep_scores<CAUSAL, TOKEN>(st.s[PAR ^ 1][0], st.s[PAR ^ 1][1], p, (t - 1) * 64, q0, qrow, hh, skt);
        full_step<D, QK_FMT, V_FMT, PAR, TWO>(st, kbuf, qbuf);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    // t = 0: QK(0) only
    {
        const unsigned char* kbuf = sync_iter(0);
        qk_chunk<QK_FMT, D>(kbuf, qbuf, st.s[0][0], st.s[0][1]);
    }
    // t = 1 .. n_w: full pipelined steps, two per trip (parity 1 then 0), no per-iteration branching
    int t = 1;
    for (; t + 1 <= n_w; t += 2) {
        full(P1{}, t);
        full(P0{}, t + 1);
    }
    if (t <= n_w) {  // n_w odd
        full(P1{}, t);
        ++t;
    }
    // t = n_w + 1: the last chunk's PV
    {
        const unsigned char* kbuf = sync_iter(t);
        if (t & 1) pv_chunk<V_FMT, D, TWO>(kbuf + CH, st.p[1], st.pl[TWO ? 1 : 0], st.o);
        else pv_chunk<V_FMT, D, TWO>(kbuf + CH, st.p[0], st.pl[0], st.o);
        ++t;
    }
    // causal: waves whose rows end earlier keep the workgroup's barrier / DMA cadence until the last wave is done
    for (; t < T; ++t) sync_iter(t);
}

// QK_FMT / V_FMT: QATTN_FMT_E4M3 (0) or QATTN_FMT_E5M2 (1) == the MFMA's cbsz/blgp selector.
template <int D, int QK_FMT, int V_FMT, bool CAUSAL, bool TOKEN, bool TWO>
__global__ __launch_bounds__(kThreads, 2) void attn_fwd_kernel_v2(const AttnParams p, const int qb_lo, const int qb_n) {
    constexpr int CH = 64 * D;      // bytes of one K (or V) chunk
    constexpr int STAGE = 2 * CH;   // K chunk + V chunk
    constexpr int KS = D / 64;      // QK^T k-steps
    constexpr int MB = D / 32;      // O^T row blocks
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hh = lane >> 5;

    // this launch covers query blocks [qb_lo, qb_lo + qb_n) of every head
    int head, qb;
    map_block(p, blockIdx.x, qb_n, CAUSAL, head, qb);
    qb += qb_lo;
    const int b = head / p.Hq, h = head % p.Hq;
    const int hkv = h / (p.Hq / p.Hkv);
    const long kv_head = (long)b * p.Hkv + hkv;
    const int q0_wg = qb * kQPerWG;
    const int q0 = q0_wg + wave * kQPerWave;  // first query row of this wave
    const int qrow = q0 + ql;                 // this lane's query row

    const unsigned char* kg = p.k + kv_head * (long)p.nchunks * CH;
    const unsigned char* vg = p.v + kv_head * (long)p.nchunks * CH;

    // chunks the workgroup / this wave must visit (causal: up to the diagonal of the last row)
    const int n_wg = CAUSAL ? min(p.nchunks, (min(q0_wg + kQPerWG, p.Sq) - 1) / 64 + 1) : p.nchunks;
    const int n_w = CAUSAL ? min(n_wg, (q0 + kQPerWave - 1) / 64 + 1) : p.nchunks;

    // start the DMA ring: stage(t) = {K chunk t, V chunk t-2}
    stage_kv<D>(kg, vg, smem, wave, lane);
    stage_kv<D>(kg + (long)min(1, p.nchunks - 1) * CH, vg, smem + STAGE, wave, lane);

    // Q^T fragments: global -> this lane's own slots of the workgroup's Q area in LDS (behind the K/V ring);
    // only the writing lane ever reads them back, so no barrier is needed (the compiler orders the lane's own
    // ds_write -> ds_read with lgkmcnt).
    unsigned char* qbuf = smem + kStagesV2 * STAGE + wave * (KS << 11) + (hh << 10) + (ql << 4);
    {
        const bool qvalid = qrow < p.Sq;
        const unsigned char* qp = p.q + (((long)b * p.Hq + h) * p.Sq + (qvalid ? qrow : 0)) * D + hh * 32;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            v4i lo = *reinterpret_cast<const v4i*>(qp + s * 64);
            v4i hi = *reinterpret_cast<const v4i*>(qp + s * 64 + 16);
            if (!qvalid) { lo = v4i{0, 0, 0, 0}; hi = v4i{0, 0, 0, 0}; }
            *reinterpret_cast<v4i*>(qbuf + (s << 11)) = lo;
            *reinterpret_cast<v4i*>(qbuf + (s << 11) + 512) = hi;
        }
    }
    // softmax scale in the exp2 domain: c = scale_q * scale_k * sm_scale * log2(e)   (tk/attention.py:204-210)
    float c;
    if (TOKEN) c = p.sm_log2e * (qrow < p.Sq ? p.sq[((long)b * p.Hq + h) * p.Sq + qrow] : 1.0f);
    else c = p.sm_log2e * p.sq[(long)b * p.Hq + h] * p.sk[kv_head];
    const float* skt = TOKEN ? p.sk + kv_head * p.Skv : nullptr;

    WaveState<D, TWO> st;
#pragma unroll
    for (int m = 0; m < MB; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) st.o[m][r] = 0.0f;
    st.m_run = -1.0e30f;  // finite sentinel: the first chunk always takes the fix-up branch
    st.l_run = 0.0f;
    st.c = c;
    kv_sweep<D, QK_FMT, V_FMT, CAUSAL, TOKEN, TWO>(st, p, smem, kg, vg, qbuf, n_wg, n_w, q0, qrow, wave, lane, skt);
    const float m_run = st.m_run, l_run = st.l_run;
    v16f (&o)[MB] = st.o;

    // ---- epilogue: combine the two half-wave partial sums, normalise, convert, store
    float l_tot;
    {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
        l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const float sv = p.sv ? p.sv[kv_head] : 1.0f;
    const float inv = sv / l_tot;
    if (qrow < p.Sq) {
        const long row_off = (((long)b * p.Hq + h) * p.Sq + qrow) * D;
        if (p.out_fmt == QATTN_FMT_BF16) {
            __bf16* op = reinterpret_cast<__bf16*>(p.out) + row_off;
#pragma unroll
            for (int m = 0; m < MB; m++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
                    bf4 t;
#pragma unroll
                    for (int i = 0; i < 4; i++) t[i] = (__bf16)(o[m][4 * j + i] * inv);
                    *reinterpret_cast<bf4*>(op + 32 * m + 8 * j + 4 * hh) = t;
                }
        } else {
            _Float16* op = reinterpret_cast<_Float16*>(p.out) + row_off;
#pragma unroll
            for (int m = 0; m < MB; m++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    h4 t;
#pragma unroll
                    for (int i = 0; i < 4; i++) t[i] = (_Float16)(o[m][4 * j + i] * inv);
                    *reinterpret_cast<h4*>(op + 32 * m + 8 * j + 4 * hh) = t;
                }
        }
        if (p.lse && hh == 0) {
            // ln sum_j exp(score_j) = ln2 * (m*c - shift) + ln(l')
            p.lse[((long)b * p.Hq + h) * p.Sq + qrow] = 0.6931471805599453f * (m_run * c - kPShift) + __logf(l_tot);
        }
    }
}

template <int D, int FMT, bool CAUSAL, bool TOKEN, bool TWO>
static int launch_attn_v2_one(const AttnParams& p, int qb_lo, int qb_n, hipStream_t st) {
    if (qb_n <= 0) return QATTN_OK;
    const int grid = p.B * p.Hq * qb_n;
    const size_t lds = (size_t)kStagesV2 * 2 * 64 * D + (size_t)kQPerWG * D;  // K/V ring + parked Q^T fragments
    auto kern = attn_fwd_kernel_v2<D, FMT, FMT, CAUSAL, TOKEN, TWO>;
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return QATTN_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, st, p, qb_lo, qb_n);
    return QATTN_OK;
}

// Query blocks whose first row sees fewer than kTwoTermKeys keys run the two-term (hi+lo fp8 P) instantiation,
// the rest the one-term one: two launches over disjoint q-block ranges of the same output tensor.
template <int D, int FMT, bool CAUSAL>
static int launch_attn_v2_t(const AttnParams& p, int scale_mode, hipStream_t st) {
    int n_two;  // leading q-blocks that need two-term P
    if (CAUSAL) n_two = min(p.nqb, ceil_div(min(kTwoTermKeys, p.Skv), kQPerWG));
    else n_two = p.Skv < kTwoTermKeys ? p.nqb : 0;
    int rc;
    if (scale_mode == QATTN_SCALE_TOKEN) {
        rc = launch_attn_v2_one<D, FMT, CAUSAL, true, false>(p, n_two, p.nqb - n_two, st);
        if (rc == QATTN_OK) rc = launch_attn_v2_one<D, FMT, CAUSAL, true, true>(p, 0, n_two, st);
    } else {
        rc = launch_attn_v2_one<D, FMT, CAUSAL, false, false>(p, n_two, p.nqb - n_two, st);
        if (rc == QATTN_OK) rc = launch_attn_v2_one<D, FMT, CAUSAL, false, true>(p, 0, n_two, st);
    }
    return rc;
}

template <int D>
static int launch_attn_v2_d(const AttnParams& p, int fmt, int causal, int scale_mode, hipStream_t st) {
    if (fmt == QATTN_FMT_E4M3) return causal ? launch_attn_v2_t<D, QATTN_FMT_E4M3, true>(p, scale_mode, st) : launch_attn_v2_t<D, QATTN_FMT_E4M3, false>(p, scale_mode, st);
    return causal ? launch_attn_v2_t<D, QATTN_FMT_E5M2, true>(p, scale_mode, st) : launch_attn_v2_t<D, QATTN_FMT_E5M2, false>(p, scale_mode, st);
}

int launch_attn_v2(const AttnParams& p, int D, int fmt, int causal, int scale_mode, hipStream_t st) {
    if (D == 128) return launch_attn_v2_d<128>(p, fmt, causal, scale_mode, st);
    return QATTN_ERR_UNSUPPORTED_DIM;
}

}  // namespace qattn
