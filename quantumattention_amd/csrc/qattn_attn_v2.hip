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
//  * LDS ring of 2*kSyncEvery+1 = 5 slots, stage(t) = {K chunk t, V chunk t-1}, filled by LDS-DMA; the waves of a
//    workgroup synchronise (vmcnt(0) + s_barrier) only every kSyncEvery = 2 iterations and then request the next two
//    stages, so the two waves of a SIMD drift apart between barriers.
//  * P is scaled by 2^kPShift before the e4m3 conversion.  One e4m3 term carries 3 mantissa bits, which is accurate enough
//    only while a row's weight is spread over many keys, so:
//      - query blocks that see fewer than kTwoTermKeys keys run with P split hi + lo (two terms, 2x the PV MFMAs);
//      - every other block runs one term and tracks two statistics per row: R = l / p_max (the inverse of the row's largest
//        softmax weight) and the effective key count l^2 / sum P'^2.  Rows that end below peak_r0 / peak_neff are "peaked"
//        -- unless the top key is the row's exact reference and the REST is provably flat (row_is_peaked, qattn_attn.h), and
//        always when the other keys average less than kCrushMean (they would sit in e4m3's subnormals):
//        up to max_rescue 32-row groups of a block are recomputed on their own (rescue_rows: two-term, split-K over the 8
//        waves); with more, the 256-row block repeats its sweep in two-term mode.  A block predicted to be peaked (score
//        moments from the pre-pass, or the spread of its first chunk of scores) starts two-term or stops its one-term sweep
//        after three chunks.  All paths live in one kernel, so a launch covers all query blocks of all heads.
//  * launches with more blocks than CUs are persistent (one workgroup per CU); causal ones, and non-causal ones with many blocks per
//    workgroup, draw their blocks from per-XCD counters (sched_next_block), heaviest first with the rescue-prone blocks ahead.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "qattn_attn.h"
#include "qattn_pv16.h"
#include "qattn_pv16p.h"

namespace qattn {

// Where the workgroup's few words live (vote words, the hand-out mailbox, the head's V chunk scale words, the Q prefetch's dump slot: 2.1 KiB).
// Behind the fp8 ring and the parked Q^T fragments -- except in the fused (Q16) kernels, whose pipelined 16-bit-V pass rings 120 KiB from
// the start of LDS: there they sit in the last 4 KiB of the CU's 160 KiB (those kernels take all of it: one workgroup per CU either way).
// The 16-bit-V rescue's parked Q^T fragments cover them too: rescue_pass and the kernel's block loop take out what they need first.
constexpr int kLdsAll = 160 * 1024;
template <int D, int NW, bool Q16>
constexpr int v2_words_offset() { return (Q16 && NW == 8) ? kLdsAll - 4096 : 5 * 2 * 64 * D + NW * kQPerWave * D; }

constexpr int kSyncEvery = 2;                   // waves synchronise every kSyncEvery 64-key iterations
constexpr int kStagesV2 = 2 * kSyncEvery + 1;   // ring slots: G live + G being filled + the previous V stage

template <int D, bool TWO, bool BYTE>
struct WaveState {
    static constexpr int MB = D / 32;
    v16f o[MB];             // O^T accumulators
    v16f s[2][2];           // S^T ping-pong: s[t&1] holds chunk t's two 32-key tiles
    v8i p[2];               // P^T (e4m3) ping-pong: p[t&1] holds chunk t
    v8i pl[TWO ? 2 : 1];    // low term of the two-term split (unused when !TWO)
    v8i vpre[2];            // V fragments (row blocks 0,1) of the NEXT iteration's PV, read one iteration ahead
    float m_run;   // reference max of the raw scores: P' = 2^(shift + (s - m_run) c); updated only by the fix-up branch
    float m_true;  // true running max (m_run <= m_true <= m_run + thr / c): gives the row's p_max for the peakedness test
    float l_run;   // this lane's partial row sum of P' (exact-exp mode)
    // BYTE mode: row sums of the quantised P', accumulated by ONE v_mfma_f32_16x16x128_f8f6f4 per chunk (32 cycles): its A
    // operand is 1.0 in row 0 for k-groups 0,2 and in row 1 for k-groups 1,3, so that with the P^T fragment as B (lane =
    // query + 32*half, 32 keys per lane) D[0][n] = sum over the 64 keys of query n and D[1][n] = that of query n + 16.
    // lsum[0] / lsum[1] of lanes 0..15 hold them; everything else in lsum stays 0.
    v4f lsum;
    // NEFF (one-term passes under QATTN_PRECISION_AUTO): row sums of P'^2 for the effective key count l^2 / sum P'^2 (DESIGN.md
    // section 4.5).  BYTE: one more row-sum MFMA per chunk on the SAME P bytes with the B format switched to e5m2 -- the exponent
    // field of an e4m3 byte weighs twice as much when read as e5m2, so the byte of 2^x reads as 0.444 .. 0.5 of 2^(2x)
    // (kNeffByteRatio; tools/models/sim_heavy.py).  Exact mode: fp32 fmas beside the row sum (l2_run).
    v4f lsq;
    float l2_run;
    v8i qreg[2];   // QREG kernels: the wave's Q^T fragments (both k-steps) held in registers instead of re-read from LDS
    v8i ones;      // BYTE mode: the all-ones A operand of that MFMA, kept opaque so it is not re-materialised every iteration
    float c;       // scale_q*scale_k*sm_scale*log2(e)
    // two values that change only with m_run (fix-up branch), kept instead of re-derived every iteration (2 + 2 VALU of ~87):
    float mcv;     // the additive constant of the exponent / byte formula for the current m_run (full_step: mc)
    float lim;     // m_run + thr / c: a row's chunk max above it means P' could overflow -> fix-up
    int vsx;       // block-scaled V: scale word (vscale_word) of the V chunk the current iteration's PV products read
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

// 4 scores -> 4 exponentials -> one dword of the e4m3 P operand (+ the residual dword when TWO); accumulates the
// partial row sums in acc[0..3] (FIRST: initialises them).  `seed` only provides the register the first
// v_cvt_pk_fp8_f32 writes its low half into (its high half is overwritten by the second), saving a v_mov.
// ACC = false (two-term passes of the kernels without an LSE output): no fp32 row sum here -- the sums of the QUANTISED terms come
// from the matrix pipe (full_step, SUMM), 36 vector instructions per chunk less in a pass that is bound by vector issue.
template <bool TWO, bool FIRST, bool NEFF = false, bool ACC = true>
__device__ __forceinline__ void exp_group(const v16f& sx, int j, float c, float mc, float (&acc)[4], v8i& pv, v8i& plv,
                                          int w, int seed, float* acc2 = nullptr) {
    float e[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        e[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[4 * j + i], c, mc));
        if (ACC) acc[i] = FIRST ? e[i] : acc[i] + e[i];
        if (NEFF) acc2[i] = FIRST ? e[i] * e[i] : __builtin_fmaf(e[i], e[i], acc2[i]);
    }
    if (ACC) asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));  // sums stay in this slot
    // (v_cvt_pk_fp8_f32 keeps the other half of its destination: handed a LIVE register as that destination -- the seed the one-term
    //  callers pass is dead where they pass it, the two-term pass's is not -- the compiler copies it first; an undefined register costs nothing)
    int fresh_hi, fresh_lo;
    asm volatile("" : "=v"(fresh_hi), "=v"(fresh_lo));
    int ph = cvt_pk_fp8<QATTN_FMT_E4M3, false>(e[0], e[1], TWO ? fresh_hi : seed);
    ph = cvt_pk_fp8<QATTN_FMT_E4M3, true>(e[2], e[3], ph);
    if (TWO) {
        // (v_cvt_pk_f32_fp8: two bytes per instruction -- the pass is bound by vector issue)
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 h01 = __builtin_amdgcn_cvt_pk_f32_fp8(ph, false), h23 = __builtin_amdgcn_cvt_pk_f32_fp8(ph, true);
        int plo = cvt_pk_fp8<QATTN_FMT_E4M3, false>(e[0] - h01[0], e[1] - h01[1], fresh_lo);
        plo = cvt_pk_fp8<QATTN_FMT_E4M3, true>(e[2] - h23[0], e[3] - h23[1], plo);
        asm volatile("" : "+v"(plo));
        plv[w] = plo;
    }
    asm volatile("" : "+v"(ph));  // keep the optimistic result materialised HERE (under the MFMA), not sunk past the fix-up branch
    pv[w] = ph;
}

// BYTE mode: 4 scores -> the e4m3 BYTES of 2^x directly.  For a normal e4m3 number byte = 8*(e+7) + m with value
// 2^e*(1+m/8), so byte ~= 8*x + 56 (Schraudolph's exponent trick at 3 mantissa bits): one fma + one saturating
// round-to-nearest v_cvt_pk_u8_f32 per score (profiles/r01_cvt_u8_probe.log) instead of fma + v_exp_f32 + half a
// v_cvt_pk_fp8_f32 (~17 issue cycles -> ~6).  c8 = 8c, off8 = 8*(shift - m*c) + 56 + kByteBias.  -inf -> 0.
__device__ __forceinline__ void byte_group(const v16f& sx, int j, float c8, float off8, v8i& pv, int w) {
    // 7 VALU per 4 scores: four v_fma_f32 (c8 / off8 carry a factor 1/65535), two v_cvt_pknorm_u16_f32 (round to nearest, clamps to
    // [0, 65535], -inf/NaN -> 0: profiles/r01_pknorm_probe.log) and one v_perm_b32 gathering the four low bytes.  (Two v_pk_fma_f32 instead
    // of the four v_fma_f32 were measured in round 1: packed-fp32 VALU stalls behind a running MFMA, +25 % cycles -- profiles/r01_ablation.md.)
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const float a0 = __builtin_fmaf(sx[4 * j + 0], c8, off8), a1 = __builtin_fmaf(sx[4 * j + 1], c8, off8);
    const float b0 = __builtin_fmaf(sx[4 * j + 2], c8, off8), b1 = __builtin_fmaf(sx[4 * j + 3], c8, off8);
    const us2 qa = __builtin_amdgcn_cvt_pknorm_u16(a0, a1), qb = __builtin_amdgcn_cvt_pknorm_u16(b0, b1);
    unsigned ua, ub;
    __builtin_memcpy(&ua, &qa, 4);
    __builtin_memcpy(&ub, &qb, 4);
    unsigned b = __builtin_amdgcn_perm(ub, ua, 0x06040200u);
    asm volatile("" : "+v"(b));  // stays in this slot
    pv[w] = (int)b;
}


#define QATTN_SLOT_FENCE() __builtin_amdgcn_sched_barrier(0)
#define QATTN_SM_GROUP(FIRST, SX, J, MC, W, SEED)                                   \
    do {                                                                            \
        if (BYTE) byte_group(SX, J, cx, MC, pc, W);                                 \
        else exp_group<TWO, FIRST, NEFF, !SUMM>(SX, J, cx, MC, acc, pc, pcl, W, SEED, acc2); \
    } while (0)

// One pipelined iteration (1 <= t <= n_w): PV(t-2), [row-sum MFMA], QK(t), softmax(t-1) in hand-placed MFMA slots.
// PAR = t & 1.  A lone wave measured ~1800 cycles per iteration when every MFMA waited for LDS fragments issued one
// short slot earlier, so the operand pipeline is two slots deep: slot i issues the ds_reads slot i+2 consumes, PV goes
// first on V fragments that were read during the PREVIOUS iteration (stage(t-1) has been visible since that barrier),
// and QK^T goes last on K fragments requested at the top of the iteration.  The softmax slices (4 scores each) sit
// under the MFMAs; only the rare fix-up waits for accumulators.  Q^T fragments are parked in LDS (registers, not LDS
// bandwidth, are the scarce resource at two waves per SIMD).
//   kbuf  : stage(t),   K part  (+ lane offset)      vprev : stage(t-1), V part = V(t-2)
//   vnext : stage(t),   V part = V(t-1) (prefetch for the next iteration)
template <int D, int QK_FMT, int V_FMT, int PAR, bool TWO, bool BYTE, int ABL = 0 /* 1024: the stamped measurement instantiation; else 0 */, bool QREG = false, bool VS = false, bool NEFF = false, bool SUMM = false, typename Stage>
__device__ __forceinline__ void full_step(WaveState<D, TWO, BYTE>& st, const unsigned char* kbuf, const unsigned char* vprev,
                                          const unsigned char* vnext, const unsigned char* qbuf, Stage&& stage, const unsigned* vx_next = nullptr) {
    static_assert(D == 128, "hand-placed slots are written for D = 128");
    auto LDSF = [&](const unsigned char* ptr) -> v8i { return lds_read_frag(ptr); };
    constexpr int PL_R = TWO ? PAR : 0, PL_W = TWO ? (PAR ^ 1) : 0;
    v16f& sn0 = st.s[PAR][0];            // S(t)   tile 0 (keys  0..31 of chunk t)
    v16f& sn1 = st.s[PAR][1];            //        tile 1 (keys 32..63)
    const v16f& sc0 = st.s[PAR ^ 1][0];  // S(t-1) tiles: the chunk being exponentiated
    const v16f& sc1 = st.s[PAR ^ 1][1];
    v8i& pc = st.p[PAR ^ 1];             // P(t-1) being produced
    v8i& pcl = st.pl[PL_W];
    const v8i& pp = st.p[PAR];           // P(t-2) consumed by PV
    const v8i& ppl = st.pl[PL_R];
    constexpr float SHIFT = BYTE ? kPShiftByte : kPShift, THR = BYTE ? kRescaleThrByte : kRescaleThr;
    const float c = st.c;
    // exact mode: p' = exp2(s*c + mc);  byte mode: byte = rne(s*c8 + mc)  (c8 = 8c, mc = 8*(shift - m*c) + 56 + bias)
    // (byte mode carries the 1/65535 of v_cvt_pknorm_u16_f32's [0,1] -> [0,65535] map in both constants)
    constexpr float U16 = 1.0f / 65535.0f;
    const float cx = BYTE ? (8.0f * U16) * c : c;
    const float mc = st.mcv;
    float acc[4], acc2[4];

    // slot 0: O0 += V0.P(t-2)            reads: V2            VALU: max over tile 0
    st.o[0] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(st.vpre[0], pp, st.o[0], st.vsx);
    if (TWO) st.o[0] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(st.vpre[0], ppl, st.o[0], st.vsx);
    v8i fc = LDSF(vprev + (2 << 11));
    // (v_max3_f32 through asm: on MFMA results the compiler otherwise adds a canonicalising v_max_f32 x, x, x per chain.  Three
    // interleaved chains: hipcc pads wait states between an asm statement and a VALU that reads its output unless two other
    // instructions sit in between, so a single chain of dependent asm v_max3 carried an s_nop per link -- eight issue slots per iteration)
    float mxa = max3_raw(sc0[0], sc0[1], sc0[2]);
    float mxb = max3_raw(sc0[3], sc0[4], sc0[5]);
    float mxc = max3_raw(sc0[6], sc0[7], sc0[8]);
    mxa = max3_raw(mxa, sc0[9], sc0[10]);
    mxb = max3_raw(mxb, sc0[11], sc0[12]);
    mxc = max3_raw(mxc, sc0[13], sc0[14]);
    QATTN_SLOT_FENCE();
    // slot 1: O1 += V1.P(t-2)            reads: V3            VALU: max over tile 1, group 0
    st.o[1] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(st.vpre[1], pp, st.o[1], st.vsx);
    if (TWO) st.o[1] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(st.vpre[1], ppl, st.o[1], st.vsx);
    v8i fd = LDSF(vprev + (3 << 11));
    mxa = max3_raw(mxa, sc0[15], sc1[0]);
    mxb = max3_raw(mxb, sc1[1], sc1[2]);
    mxc = max3_raw(mxc, sc1[3], sc1[4]);
    mxa = max3_raw(mxa, sc1[5], sc1[6]);
    mxb = max3_raw(mxb, sc1[7], sc1[8]);
    mxc = max3_raw(mxc, sc1[9], sc1[10]);
    mxa = max3_raw(mxa, sc1[11], sc1[12]);
    mxb = max3_raw(mxb, sc1[13], sc1[14]);
    mxc = max3_raw(mxc, sc1[15], sc1[15]);
    float mx = max3_raw(mxa, mxb, mxc);
    QATTN_SM_GROUP(true, sc0, 0, mc, 0, pp[0]);
    QATTN_SLOT_FENCE();
    // slot 2: O2 += V2.P(t-2)            reads: Q k-step 0, K(tile 0, k-step 0)      VALU: group 1
    st.o[2] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(fc, pp, st.o[2], st.vsx);
    if (TWO) st.o[2] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(fc, ppl, st.o[2], st.vsx);
    v8i qf;
    if (QREG) qf = st.qreg[0]; else qf = LDSF(qbuf);
    v8i ka = LDSF(kbuf + (0 << 11));
    QATTN_SM_GROUP(false, sc0, 1, mc, 1, pc[0]);
    QATTN_SLOT_FENCE();
    // slot 3: O3 += V3.P(t-2)            reads: K(tile 1, k-step 0)                  VALU: group 2
    st.o[3] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(fd, pp, st.o[3], st.vsx);
    if (TWO) st.o[3] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(fd, ppl, st.o[3], st.vsx);
    v8i kb = LDSF(kbuf + (2 << 11));
    QATTN_SM_GROUP(false, sc0, 2, mc, 2, pc[1]);
    QATTN_SLOT_FENCE();
    stage();  // K/V staging of a later chunk: after the PV slots are in flight, not between the barrier and the first MFMA
    // slot 4 (BYTE): row sum of the quantised P(t-2) on the matrix pipe: ones(32x64).P^T -> every row = sum over 64 keys
    if (BYTE || SUMM) st.lsum = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(st.ones, pp, st.lsum, QATTN_FMT_E4M3, QATTN_FMT_E4M3, 0, 0, 0, 0);
    if (SUMM && TWO) st.lsum = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(st.ones, ppl, st.lsum, QATTN_FMT_E4M3, QATTN_FMT_E4M3, 0, 0, 0, 0);
    // ... and of its bytes read as e5m2 ~= P'^2 / 2 (WaveState::lsq)
    if (BYTE && NEFF) st.lsq = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(st.ones, pp, st.lsq, QATTN_FMT_E4M3, QATTN_FMT_E5M2, 0, 0, 0, 0);
    v8i qg;                            // Q k-step 1
    if (QREG) qg = st.qreg[1]; else qg = LDSF(qbuf + (1 << 11));
    v8i kc = LDSF(kbuf + (1 << 11));   // K(tile 0, k-step 1)
    QATTN_SM_GROUP(false, sc0, 3, mc, 3, pc[2]);
    QATTN_SLOT_FENCE();
    // slot 5: S0 = K(0,0).Q0             reads: K(tile 1, k-step 1)                  VALU: group 4
#pragma unroll
    for (int r = 0; r < 16; r++) { sn0[r] = 0.0f; sn1[r] = 0.0f; }
    sn0 = mfma_f8<QK_FMT, QK_FMT>(ka, qf, sn0);
    v8i kd = LDSF(kbuf + (3 << 11));
    QATTN_SM_GROUP(false, sc1, 0, mc, 4, pc[3]);
    QATTN_SLOT_FENCE();
    // slot 6: S1 = K(1,0).Q0             reads: next iteration's V0                  VALU: group 5
    sn1 = mfma_f8<QK_FMT, QK_FMT>(kb, qf, sn1);
    st.vpre[0] = LDSF(vnext + (0 << 11));
    // block-scaled V: the scale byte of the NEXT iteration's PV products, requested here (this iteration's four are issued) -- read
    // at the top of its own iteration it put a whole LDS round trip in front of the first MFMA of every iteration
    if (VS) st.vsx = (int)*vx_next;
    QATTN_SM_GROUP(false, sc1, 1, mc, 5, pc[4]);
    QATTN_SLOT_FENCE();
    // slot 7: S0 += K(0,1).Q1            reads: next iteration's V1                  VALU: group 6
    sn0 = mfma_f8<QK_FMT, QK_FMT>(kc, qg, sn0);
    st.vpre[1] = LDSF(vnext + (1 << 11));
    QATTN_SM_GROUP(false, sc1, 2, mc, 6, pc[5]);
    QATTN_SLOT_FENCE();
    // slot 8: S1 += K(1,1).Q1                                                        VALU: group 7, max exchange
    sn1 = mfma_f8<QK_FMT, QK_FMT>(kd, qg, sn1);
    QATTN_SM_GROUP(false, sc1, 3, mc, 7, pc[6]);
    {
        // (v_max3 through asm: fmaxf on these would be preceded by two canonicalising v_max_f32 x, x, x)
        // (the two chains meet here; m_true takes the exchanged halves directly: two independent asm statements, no pad)
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        if (!TWO) st.m_true = max3_raw(st.m_true, __uint_as_float(sw[0]), __uint_as_float(sw[1]));
        mx = max3_raw(__uint_as_float(sw[0]), __uint_as_float(sw[1]), __uint_as_float(sw[1]));
    }
    float ls = (BYTE || SUMM) ? 0.0f : (acc[0] + acc[1]) + (acc[2] + acc[3]);
    float ls2 = (BYTE || !NEFF) ? 0.0f : (acc2[0] + acc2[1]) + (acc2[2] + acc2[3]);
    QATTN_SLOT_FENCE();
    // rare fix-up: some row's max grew by more than the threshold (always on the first chunk: m_run = -1e30):
    // rescale everything accumulated so far (O and the row sum include chunk t-2) and redo this chunk's exponentials
    if (__builtin_expect(__any(mx > st.lim) != 0, 0)) {   // (mx - m_run) c > THR
        const float m_new = fmaxf(st.m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((st.m_run - m_new) * c);
#pragma unroll
        for (int m = 0; m < D / 32; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) st.o[m][r] *= alpha;
        st.l_run *= alpha;
        if (NEFF) st.l2_run *= alpha * alpha;
        if (BYTE || SUMM) {
            // lane n < 16 holds the sums of queries n (its own alpha) and n + 16 (lane n+16's alpha)
            const float alpha16 = __uint_as_float(swizzle_xor16(__float_as_uint(alpha)));   // (lanes 0..15 get lanes 16..31, no lane-index register)
            st.lsum[0] *= alpha;
            st.lsum[1] *= alpha16;
            if (NEFF) { st.lsq[0] *= alpha * alpha; st.lsq[1] *= alpha16 * alpha16; }
        }
        st.m_run = m_new;
        const float mc2 = BYTE ? __builtin_fmaf((-8.0f * U16) * m_new, c, (8.0f * SHIFT + 56.0f + kByteBias) * U16) : SHIFT - m_new * c;
        st.mcv = mc2;
        st.lim = m_new + THR / c;
        QATTN_SM_GROUP(true, sc0, 0, mc2, 0, 0);
#pragma unroll
        for (int j = 1; j < 4; j++) QATTN_SM_GROUP(false, sc0, j, mc2, j, 0);
#pragma unroll
        for (int j = 0; j < 4; j++) QATTN_SM_GROUP(false, sc1, j, mc2, 4 + j, 0);
        ls = (BYTE || SUMM) ? 0.0f : (acc[0] + acc[1]) + (acc[2] + acc[3]);
        ls2 = (BYTE || !NEFF) ? 0.0f : (acc2[0] + acc2[1]) + (acc2[2] + acc2[3]);
    }
    st.l_run += ls;
    if (NEFF) st.l2_run += ls2;
}

// One stage of the K/V ring by LDS-DMA (8-wave workgroups: one 1 KiB piece of K and one of V per wave): K at byte offset koff of
// the head, V at voff, into the ring slot at lds_off.  kv_sweep issues stages in order; block_pass issues the first kSyncEvery of
// them itself, together with the block's other loads (first_stages_issued).
template <int D>
__device__ __forceinline__ void stage_dma8(const unsigned char* kg, const unsigned char* vg, unsigned koff, unsigned voff, unsigned char* smem,
                                           unsigned lds_off, int wave, int lane) {
    constexpr int CH = 64 * D;
    const unsigned piece = ((unsigned)wave << 10) + ((unsigned)lane << 4);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kg + (koff + piece)),
                                     (__attribute__((address_space(3))) void*)(smem + lds_off + (wave << 10)), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vg + (voff + piece)),
                                     (__attribute__((address_space(3))) void*)(smem + lds_off + CH + (wave << 10)), 16, 0, 0);
}

// The KV sweep of one wave.  Returns false with st.o / st.l_run / st.m_run final.
// `forecast` (one-term byte-exponential passes under QATTN_PRECISION_AUTO): every wave measures the variance of its 32 x 64
// scores of the first chunk and predicts the smallest R = l / p_max of its rows from it (predicted_r with kPeakZWide: 2048
// samples, so the measured spread includes what an isotropic moment estimate misses -- a few large dimensions in q and k --
// and callers without moments get it too); when more waves than max_rescue predict a row below the threshold, the block is
// going to be repeated in two-term mode anyway: after two more chunks (the votes travel through the sweep's own barrier) all
// waves drain the ring, stop and return true -- 3 of n chunks wasted instead of all of them.  Chunk 0 stands for the whole
// key range here; where it does not, the R test at the end of the sweep is still the arbiter.
template <int D, int NW, int QK_FMT, int V_FMT, bool CAUSAL, bool TOKEN, bool TWO, bool BYTE, int ABL = 0, bool QREG = false, bool VS = false, bool NEFF = false, bool SUMM = false, typename LoadQ>
__device__ __forceinline__ bool kv_sweep(WaveState<D, TWO, BYTE>& st, const AttnParams& p, unsigned char* smem,
                                         const unsigned char* kg, const unsigned char* vg, const unsigned char* qbuf, int n_wg,
                                         int n_w, int q0, int qrow, int wave, int lane, const float* skt, LoadQ&& load_q,
                                         bool forecast, unsigned* vote, const unsigned* vx,   // vx (VS): the V chunks' scale bytes in LDS
                                         bool first_stages_issued) {   // first_stages_issued: the caller has requested stages 0 .. kSyncEvery - 1 already
    constexpr int CH = 64 * D, STAGE = 2 * CH;
    const int hh = lane >> 5;
    const int T = n_wg + 2;  // iterations t = 0 .. n_wg+1 : QK(t), softmax(t-1), PV(t-2)
    const int frag_lane_off = (hh << 10) + ((lane & 31) << 4);
#pragma unroll
    for (int w = 0; w < 8; w++) {
        st.p[0][w] = 0; st.p[1][w] = 0;
        st.pl[0][w] = 0;
        if (TWO) st.pl[TWO ? 1 : 0][w] = 0;
    }
    // Ring protocol: stage(t) = {K(t), V(t-1)} lives in slot t % kStagesV2 and is filled by LDS-DMA.  Waves synchronise
    // only every G = kSyncEvery iterations: barrier(t), t % G == 0, publishes stages t .. t+G-1 (DMA issued at iteration
    // t-G, waited with vmcnt(0) just before the barrier), after which every wave issues the DMA of stages t+G .. t+2G-1.
    // Between barriers a wave runs G 64-key iterations freely, so the two waves of a SIMD drift apart instead of being
    // re-aligned every chunk.  Slots live during iterations t .. t+G-1: stages t-1 (V of PV) .. t+G-1; being written:
    // t+G .. t+2G-1  ->  2G+1 distinct slots.
    // Stages are issued strictly in order, so the DMA source/destination advance incrementally (a handful of SALU per
    // stage instead of ~40 for the modulo / clamp / 64-bit address arithmetic of an indexed form): koff = byte offset of
    // K(min(t, n-1)) within the head, voff = that of V(min(max(t-1, 0), n-1)) = the previous stage's koff.
    static_assert(NW == 8 && 2 * 64 * D / (NW * 1024) == 2, "one K and one V DMA per wave and stage");
    const unsigned char* kg_w = kg + (wave << 10);
    const unsigned char* vg_w = vg + (wave << 10);
    const unsigned koff_max = (unsigned)(p.nchunks - 1) * CH;
    unsigned koff = 0, voff = 0, lds_next = 0;
    const unsigned lane16 = (unsigned)lane << 4;
    auto dma_next = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kg_w + (koff + lane16)),
                                         (__attribute__((address_space(3))) void*)(smem + lds_next + (wave << 10)), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vg_w + (voff + lane16)),
                                         (__attribute__((address_space(3))) void*)(smem + lds_next + CH + (wave << 10)), 16, 0, 0);
        voff = koff;
        koff = min(koff + (unsigned)CH, koff_max);
        lds_next = lds_next + STAGE == kStagesV2 * STAGE ? 0u : lds_next + STAGE;
    };
    auto dma_for = [&](int) { dma_next(); };
    static_assert(kSyncEvery == 2, "block_pass requests two stages");
    if (first_stages_issued) {   // (workgroup-uniform; T >= 3) the state dma_next would have left behind
        voff = min((unsigned)CH, koff_max);
        koff = min(voff + (unsigned)CH, koff_max);
        lds_next = 2 * STAGE;
    } else {
#pragma unroll
        for (int g = 0; g < kSyncEvery; g++)
            if (g < T) dma_for(g);
    }
    load_q();  // the wave's Q^T rows (global -> [quantise ->] LDS / registers) travel while the first K/V stages do
    unsigned slot_cur = 0, slot_prev = 0;
    auto sync_iter = [&](int t, bool in_step = false) __attribute__((always_inline)) -> const unsigned char* {
        (void)in_step;
        if (t % kSyncEvery == 0) {
            wait_vmcnt<0>();  // this wave's pieces of stages t .. t+G-1 have landed
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int g = 0; g < kSyncEvery; g++)
                if (t + kSyncEvery + g < T) dma_for(t + kSyncEvery + g);
        } else {
            asm volatile("s_nop 0" ::: "memory");  // keeps the iterations of a group separate scheduling regions
        }
        return smem + slot_cur + frag_lane_off;
    };
    auto advance = [&]() __attribute__((always_inline)) {  // slot_cur / slot_prev: LDS offsets of stage(t) / stage(t-1), advanced once per iteration
        slot_prev = slot_cur;
        slot_cur = slot_cur + STAGE == kStagesV2 * STAGE ? 0u : slot_cur + STAGE;
    };
    auto do_stage = [&](int) {};
    // (VS) the scale byte of V(t - 1), requested during iteration t: a running LDS address (one scalar add per iteration; the table
    // holds nchunks <= kVxWords live entries, a longer head has no chunk scales and re-reads entry 0 = 2^0)
    const unsigned* vx_next = vx;
    const int vx_step = (VS && p.vexp != nullptr) ? 4 : 0;
    // ragged_tag: may chunk t - 1 (the one this iteration exponentiates) reach past the key range?  Only a head's last chunk can;
    // non-causal sweeps say so statically for all iterations but the last (prep_scores)
    auto full = [&](auto par_tag, int t, auto ragged_tag) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_tag)::value;
        constexpr bool RAGGED = decltype(ragged_tag)::value;
        const unsigned char* kbuf;
        kbuf = sync_iter(t, true);
        const unsigned char* vprev = smem + slot_prev + CH + frag_lane_off;
        advance();
        // (RAGGED = false, head-wise: the chunk lies inside the key range AND below the wave's causal diagonal -- nothing to prepare)
        if constexpr (RAGGED || TOKEN) prep_scores<CAUSAL, TOKEN, RAGGED>(st.s[PAR ^ 1][0], st.s[PAR ^ 1][1], p, (t - 1) * 64, q0, qrow, hh, skt);
        auto stage = [&]() { do_stage(t); };
        // (VS) iteration t + 1 multiplies V(t - 1): its scale byte is requested during iteration t (t = 1 runs on the initial 2^0: P = 0)
        full_step<D, QK_FMT, V_FMT, PAR, TWO, BYTE, ABL, QREG, VS, NEFF, SUMM>(st, kbuf, vprev, kbuf + CH, qbuf, stage, vx_next);
        if constexpr (VS) vx_next = reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned char*>(vx_next) + vx_step);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    // An iteration of the main loop exponentiates a chunk <= n_w - 2.  Non-causal: not the head's last chunk, hence inside the key range.
    // Causal: the wave's rows start at q0 = 32 j and n_w = (q0 + 31) / 64 + 1, so chunk n_w - 2 ends at key 64 (n_w - 1) - 1 <= q0 - 1 for
    // even j and <= q0 - 33 for odd j: at or below the diagonal of every row of the wave -- only the wave's LAST chunk meets the mask
    // (and the ragged tail), and that one runs in the copies with the test.  Round 3 kept the run-time test in the causal loop: 345
    // instead of 276 instructions per two iterations, 5.1 scalar instructions per MFMA at C3 (VERDICT r3 Weak-5).
    using Inner = std::integral_constant<bool, false>;
    using Last = std::integral_constant<bool, true>;
    if (BYTE || SUMM) {
        {   // A of the row-sum MFMA: lane = row (l & 15) + 16 * k-group; rows 0 / 1 are 1.0 (e4m3 0x38) on even / odd k-groups
            const int row = lane & 15, kg = lane >> 4;
            const int one = ((row == 0 && !(kg & 1)) || (row == 1 && (kg & 1))) ? 0x38383838 : 0;
#pragma unroll
            for (int w = 0; w < 8; w++) st.ones[w] = one;
        }
    }

    // t = 0: QK(0) only
    {
        const unsigned char* kbuf = sync_iter(0);
        qk_chunk<QK_FMT, D>(kbuf, qbuf, st.s[0][0], st.s[0][1]);
        st.vpre[0] = lds_read_frag(kbuf + CH + (0 << 11));  // stage(0)'s V part (= V(0), multiplied by P = 0 at t = 1)
        st.vpre[1] = lds_read_frag(kbuf + CH + (1 << 11));
        advance();
    }
    // The row's reference starts at chunk 0's maximum.  Left at the -1e30 sentinel, the first full step always took the fix-up
    // branch: 64 multiplications of an all-zero O^T by alpha = 0 and a second pass over the chunk's exponentials, with every wave of
    // the workgroup in it at once (nothing on the matrix pipe meanwhile) -- once per block.  Setting m_run / mcv / lim here with the
    // branch's own expressions gives the same bits: the first step's optimistic exponentials ARE the branch's recomputed ones.
    if constexpr (!TOKEN) {   // (token-wise key scales are applied by prep_scores, which must then run once per chunk)
        constexpr float SHIFT = BYTE ? kPShiftByte : kPShift, THR = BYTE ? kRescaleThrByte : kRescaleThr;
        constexpr float U16 = 1.0f / 65535.0f;
        prep_scores<CAUSAL, TOKEN, true>(st.s[0][0], st.s[0][1], p, 0, q0, qrow, hh, skt);   // (idempotent: the first step masks chunk 0 again)
        float mx0 = max32_after_mfma(st.s[0][0], st.s[0][1]);
        const auto sw0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx0), __float_as_uint(mx0), false, false);
        mx0 = fmaxf(__uint_as_float(sw0[0]), __uint_as_float(sw0[1]));
        const float m_new = fmaxf(st.m_run, mx0);
        st.m_run = m_new;
        st.mcv = BYTE ? __builtin_fmaf((-8.0f * U16) * m_new, st.c, (8.0f * SHIFT + 56.0f + kByteBias) * U16) : SHIFT - m_new * st.c;
        st.lim = m_new + THR / st.c;
    }
    // t = 1 .. n_w: full pipelined steps, two per trip (parity 1 then 0), no per-iteration branching
    int t = 1;
    constexpr bool FORECAST = !TOKEN && !TWO && BYTE;   // (run-time: only passes that check their rows ask for it)
    // every wave of the workgroup runs the first two iterations (causal: wave 0 has the fewest chunks) and the sweep is long enough to matter
    const int n_w0 = CAUSAL ? min(n_wg, (q0 - wave * kQPerWave + kQPerWave - 1) / 64 + 1) : n_w;
    forecast = FORECAST && forecast && n_w0 >= 3 && n_wg >= 16;
    if constexpr (FORECAST) {
        if (forecast) {
            // spread of the wave's 32 x 64 scores of chunk 0 (all visible: causal blocks past the first), in natural-log units
            float su = 0.0f, sq = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                su += st.s[0][0][r] + st.s[0][1][r];
                sq = __builtin_fmaf(st.s[0][0][r], st.s[0][0][r], __builtin_fmaf(st.s[0][1][r], st.s[0][1][r], sq));
            }
            su = wave_allsum(su);
            sq = wave_allsum(sq);
            const float mean = su * (1.0f / 2048.0f), cn = st.c * 0.6931471805599453f;
            const float var = fmaxf(sq * (1.0f / 2048.0f) - mean * mean, 0.0f) * cn * cn;
            const int nkeys = CAUSAL ? min(p.Skv, q0 + kQPerWave) : p.Skv;
            // (below the dead band the block already started in the right mode: borderline causal rows are cheaper to rescue)
            const bool mine = var >= kVarDeadband && many_rows_peaked((float)nkeys, var);   // the same value in every lane
            if (lane == 0) vote[wave] = mine ? 1u : 0u;
        }
    }
    // iterations t <= n_w - 1 exponentiate chunks <= n_w - 2; the last one or two iterations (the head's last chunk is among
    // them for a wave that sees it) run the copies with the ragged-tail test.  (Loop + loop + if, all with static parities:
    // an if / else-if over the leftover count made the compiler keep the S / P arrays in scratch memory.)
    for (; t + 1 <= n_w - 1; t += 2) {
        full(P1{}, t, Inner{});
        full(P0{}, t + 1, Inner{});   // (t = 1: starts with the barrier that publishes the forecast votes)
        if constexpr (FORECAST) {
            // the votes are read after the first pair of iterations (forecast: n_w >= 3 and n_wg >= 16, so this trip exists for every
            // wave of the workgroup); one scalar test per trip instead of a peeled copy of the pair, which cost registers
            if (forecast && t == 1) {
                int nf = 0;
#pragma unroll
                for (int w = 0; w < NW; w++) nf += vote[w] != 0u ? 1 : 0;
                if (__builtin_amdgcn_readfirstlane(nf) * 2 > NW) {   // most waves expect more peaked rows than a rescue is worth
                    wait_vmcnt<0>();   // the stages requested at that barrier
                    __builtin_amdgcn_s_barrier();
                    return true;
                }
            }
        }
    }
    for (; t + 1 <= n_w; t += 2) {  // at most one trip
        full(P1{}, t, Last{});
        full(P0{}, t + 1, Last{});
    }
    if (t <= n_w) {  // n_w odd
        full(P1{}, t, Last{});
        ++t;
    }
    // t = n_w + 1: the last chunk's PV (V(t-2) lives in stage(t-1); its row blocks 0,1 are already in vpre)
    {
        (void)sync_iter(t);
        const unsigned char* vprev = smem + slot_prev + CH + frag_lane_off;
        const v8i fc = lds_read_frag(vprev + (2 << 11)), fd = lds_read_frag(vprev + (3 << 11));
        // (VS: st.vsx already holds the scale of V(t - 2) = V(n_w - 1), requested by the last full step)
        // two fully static copies: any run-time choice between st.p[0] and st.p[1] (even by value) ends up as a pointer
        // phi that keeps the P registers in scratch memory
        auto tail = [&](auto par_tag) {
            constexpr int PAR = decltype(par_tag)::value;
            const v8i& pp = st.p[PAR];
            const v8i& ppl = st.pl[TWO ? PAR : 0];
            st.o[0] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(st.vpre[0], pp, st.o[0], st.vsx);
            st.o[1] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(st.vpre[1], pp, st.o[1], st.vsx);
            st.o[2] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(fc, pp, st.o[2], st.vsx);
            st.o[3] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(fd, pp, st.o[3], st.vsx);
            if (TWO) {
                st.o[0] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(st.vpre[0], ppl, st.o[0], st.vsx);
                st.o[1] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(st.vpre[1], ppl, st.o[1], st.vsx);
                st.o[2] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(fc, ppl, st.o[2], st.vsx);
                st.o[3] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(fd, ppl, st.o[3], st.vsx);
            }
            if (BYTE || SUMM) st.lsum = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(st.ones, pp, st.lsum, QATTN_FMT_E4M3, QATTN_FMT_E4M3, 0, 0, 0, 0);
            if (SUMM && TWO) st.lsum = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(st.ones, ppl, st.lsum, QATTN_FMT_E4M3, QATTN_FMT_E4M3, 0, 0, 0, 0);
            if (BYTE && NEFF) st.lsq = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(st.ones, pp, st.lsq, QATTN_FMT_E4M3, QATTN_FMT_E5M2, 0, 0, 0, 0);
        };
        if (t & 1) tail(P1{}); else tail(P0{});
        ++t;
    }
    // causal: waves whose rows end earlier keep the workgroup's barrier / DMA cadence until the last wave is done
    for (; t < T; ++t) sync_iter(t);
    return false;
}

// The next block of a dynamic launch is requested when the block's KV sweep is over, BEFORE its rows are normalised and stored, and
// taken delivery of behind those stores (thread 0; `mail`: the LDS word the workgroup reads after the block's end barrier, nullptr:
// static launch or not the block's first pass).  Requested after the stores -- round 3 -- the atomic sat behind their write
// acknowledgements (one counter) with every wave waiting at the hand-over barrier: 1.2 .. 2 us per block in the dev work log of a C3
// launch.  Requested in the block's PROLOGUE (measured this round, profiles/r04/ab_c3_draw_in_prologue_dropped.log) the round trip
// is free but a workgroup then holds a reserved block for the length of a whole block: C3 +1.4 %, C5 +1.6 % (idle time before the end
// of a C3 launch 21 -> 35 us).
__device__ __forceinline__ unsigned draw_issue(const AttnParams& p, volatile unsigned* mail, int tid) {
    unsigned ticket = 0u;
    if (mail != nullptr && tid == 0) ticket = sched_draw_issue(p.sched, (int)blockIdx.x & (p.sched_nq - 1));
    return ticket;
}
__device__ __forceinline__ void draw_finish(const AttnParams& p, volatile unsigned* mail, int tid, unsigned ticket) {
    if (mail != nullptr && tid == 0) {
        const int nq = p.sched_nq;
        lds_write_word_raw(mail, (unsigned)sched_draw_finish(p.sched, nq, (int)blockIdx.x & (nq - 1), p.total_blocks / nq, (int)gridDim.x / nq, ticket));
    }
}

// One pass of a wave over its KV range with P in TWO (hi + lo) or one term, BYTE-exponential or exact, followed by the
// row sums.  With `check_peaked`, rows of the WORKGROUP that turned out peaked (largest softmax weight 1 / R above
// 1 / peak_r0) make it return, workgroup-uniform: kPassRedo -- too many, nothing of the flagged waves is stored, the
// caller repeats the block in two-term mode -- or the bit mask of the (at most max_rescue) waves whose 32-row groups
// rescue_pass then recomputes; the other waves' rows (and the optional LSE) are stored.  0: everything is stored.
constexpr int kPassRedo = 1 << 30;
template <int D, int NW, int QK_FMT, int V_FMT, bool CAUSAL, bool TOKEN, bool TWO, bool BYTE, int ABL, bool QREG, bool VS = false, bool NEFF = false, bool SUMM = false, typename LoadQ>
__device__ __forceinline__ int attend_block(const AttnParams& p, unsigned char* smem, const unsigned char* kg, const unsigned char* vg,
                                             const unsigned char* qbuf, unsigned* vote, int n_wg, int n_w, int q0, int qrow, int wave,
                                             int lane, long bh, long kv_head, float c, const float* skt, bool check_peaked, LoadQ&& load_q,
                                             const unsigned* vx, bool first_stages_issued, volatile unsigned* mail, long o_head) {   // o_head: out_head_offset(p, b, h)
    constexpr int MB = D / 32;
    const int hh = lane >> 5;
    WaveState<D, TWO, BYTE> st;
#pragma unroll
    for (int r = 0; r < 4; r++) { st.lsum[r] = 0.0f; st.lsq[r] = 0.0f; }
    st.l2_run = 0.0f;
#pragma unroll
    for (int m = 0; m < MB; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) st.o[m][r] = 0.0f;
    st.m_run = -1.0e30f;  // finite sentinel: the first chunk always takes the fix-up branch
    st.m_true = -1.0e30f;
    st.l_run = 0.0f;
    st.c = c;
    st.mcv = 0.0f;       // (the first chunk always takes the fix-up branch, which sets both)
    st.lim = -1.0e30f;
    st.vsx = kScaleWordOne;
    // ABL & 1024: the MEASUREMENT instantiation (qattn_fp8_quant_attention_forward_stamped): every wave brackets its KV sweep with
    // the shader-cycle counter and the 100 MHz real-time counter; their ratio is the clock the chip held INSIDE the kernel
    // (MI355X_MICROARCH.md, DVFS give-back item 6).  The stamps go to a buffer of their own, nothing is computed from them; the
    // product instantiations (ABL = 0) execute no stamp.
    unsigned long long stamp_t0 = 0, stamp_r0 = 0;
    if constexpr ((ABL & 1024) != 0) {
        stamp_t0 = __builtin_amdgcn_s_memtime();
        stamp_r0 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) alone: the sweep's own LDS waits stay counted
    }
    // the Q^T rows travel to LDS while the first K/V stages do; QREG kernels then keep the fragments in registers
    auto load_q_frags = [&]() {
        load_q();
        if (QREG) {
            st.qreg[0] = lds_read_frag(qbuf);
            st.qreg[1] = lds_read_frag(qbuf + (1 << 11));
        }
    };
    if (kv_sweep<D, NW, QK_FMT, V_FMT, CAUSAL, TOKEN, TWO, BYTE, ABL, QREG, VS, NEFF, SUMM>(st, p, smem, kg, vg, qbuf, n_wg, n_w, q0, qrow, wave, lane, skt, load_q_frags,
                                                                                !TWO && check_peaked, vote, vx, first_stages_issued))
    {   // forecast: the block is peaked, nothing was stored (its successor is drawn here: the repeated pass draws nothing)
        const int t_ = (wave << 6) | lane;
        draw_finish(p, mail, t_, draw_issue(p, mail, t_));
        return kPassRedo;
    }
    if constexpr ((ABL & 1024) != 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && p.stamp_buf) {
            const long wid = (bh * p.nqb + (q0 - wave * kQPerWave) / (NW * kQPerWave)) * NW + wave;
            p.stamp_buf[2 * wid] = t1 - stamp_t0;
            p.stamp_buf[2 * wid + 1] = r1 - stamp_r0;
        }
    }
    const float m_run = st.m_run, l_run = st.l_run;
    v16f (&o)[MB] = st.o;
    const int tid_draw = (wave << 6) | lane;
    const unsigned ticket = draw_issue(p, mail, tid_draw);   // (a forecast exit above draws in the repeated pass instead)

    // ---- combine the two half-wave partial sums
    float l_tot, l2_tot = 0.0f;   // l2_tot: sum of P'^2 (exact mode) or kNeffByteRatio of it (BYTE)
    if (BYTE || SUMM) {
        // query q's sum sits in lane q & 15, register q >> 4 (both half-waves' keys already added by the MFMA)
        const float s0 = bcast_low16(st.lsum[0]), s1 = bcast_low16(st.lsum[1]);
        l_tot = (lane & 16) ? s1 : s0;
        if (NEFF) {
            const float t0 = bcast_low16(st.lsq[0]), t1 = bcast_low16(st.lsq[1]);
            l2_tot = (lane & 16) ? t1 : t0;
        }
    } else {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
        l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        if (NEFF) {
            auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(st.l2_run), __float_as_uint(st.l2_run), false, false);
            l2_tot = __uint_as_float(sw2[0]) + __uint_as_float(sw2[1]);
        }
    }
    constexpr float SHIFT = BYTE ? kPShiftByte : kPShift;
    if (!TWO && check_peaked) {  // workgroup-uniform
        // R = l' / p'_max with p'_max = 2^(shift + (m_true - m_run) c): the inverse of the row's largest softmax weight
        const float r_inv_pmax = __builtin_amdgcn_exp2f(-(SHIFT + (st.m_true - m_run) * c));
        const bool peaked = qrow < p.Sq && row_is_peaked<BYTE, NEFF>(p, l_tot, l2_tot, r_inv_pmax, st.m_true == m_run, (float)(CAUSAL ? min(qrow + 1, p.Skv) : p.Skv));
        // per-wave masks of the peaked ROWS (both half-waves reach the same verdict: bit r = the wave's row r) -> after the barrier every
        // wave knows how many rows of the block need help.  Few: they are gathered across the waves into dense 32-row groups and
        // recomputed (rescue_pass); many: the block repeats in two-term mode.  Rows, not 32-row groups, are the unit: on data with a
        // score spread of 1.2 .. 1.6 about 1 .. 6 % of the rows end peaked -- a handful per block, but spread over most of its eight
        // groups, which used to send nearly every such block into the two-term repeat.
        const unsigned mine = (unsigned)__ballot(peaked);   // (low word: lanes 0 .. 31)
        const bool keep = !peaked;
        // every row that is final is stored right away (under the other waves' last iterations, as in the unchecked kernel); the
        // peaked ones are left to whoever recomputes them -- a second store to the same address from another wave is not ordered
        // behind this one
        const float sv = p.sv ? scalar_load_f32(p.sv + kv_head) : 1.0f;
        store_o_rows<MB>(p.out, p.out_fmt, o, sv / l_tot, out_row_offset(p, o_head, bh, qrow, MB * 64), hh, qrow < p.Sq && keep);
        if (p.lse && hh == 0 && qrow < p.Sq && keep)
            p.lse[bh * p.lse_stride + qrow] = (0.6931471805599453f * (m_run * c - SHIFT) + __logf(l_tot) - (BYTE ? kByteLseBias : 0.0f)) * p.lse_mul;
        draw_finish(p, mail, tid_draw, ticket);
        static_assert(NW <= 8, "eight vote words");
        // (bytes 40 .. 47 behind the vote words and the mailbox: one per wave, "a flagged row of mine is severely peaked" -- kPeakR16)
        const bool severe_mine = __any(peaked && l_tot * r_inv_pmax < kPeakR16) != 0;
        if (lane == 0) {
            lds_write_word_raw(vote + wave, mine);
            lds_write_byte_raw(reinterpret_cast<volatile unsigned char*>(vote + 10) + wave, severe_mine ? 1u : 0u);
        }
        lds_barrier();   // also: every wave is done with the K/V ring (the O rows just stored are nobody else's business)
        int nrows = 0;
        {
            v4i va, vb;
            lds_read_8words_raw(vote, va, vb);
#pragma unroll
            for (int w = 0; w < NW; w++) nrows += __builtin_popcount((unsigned)(w < 4 ? va[w & 3] : vb[w & 3]));
        }
        nrows = __builtin_amdgcn_readfirstlane(nrows);
        if (nrows == 0) return 0;
        if constexpr (!TOKEN && NW == 8) {
            if (nrows <= p.max_rescue_rows) {   // rescue_pass gathers them from the vote words
                const unsigned sev = lds_read_word_raw(vote + 10) | lds_read_word_raw(vote + 11);
                return nrows | (__builtin_amdgcn_readfirstlane((int)sev) != 0 ? kRescueSevere : 0);
            }
        }
        return kPassRedo;   // many peaked rows: the whole block repeats in two-term mode (and rewrites every row)
    }

    // ---- normalise, convert, store
    const float sv = p.sv ? scalar_load_f32(p.sv + kv_head) : 1.0f;
    const float inv = sv / l_tot;
    store_o_rows<MB>(p.out, p.out_fmt, o, inv, out_row_offset(p, o_head, bh, qrow, MB * 64), hh, qrow < p.Sq);
    draw_finish(p, mail, tid_draw, ticket);
    if (qrow < p.Sq) {
        if (p.lse && hh == 0) {
            // ln sum_j exp(score_j) = ln2 * (m*c - shift) + ln(l'); QATTN_LSE_REFERENCE: the reference's (disabled) vector
            // -(ln l + m ln2) sqrt(D) in rows padded to 16 bytes (tk/attention.py:333-346, 439-446)
            const float lse = 0.6931471805599453f * (m_run * c - SHIFT) + __logf(l_tot) - (BYTE ? kByteLseBias : 0.0f);
            p.lse[bh * p.lse_stride + qrow] = lse * p.lse_mul;
        }
        if (TWO && p.path && hh == 0) p.path[bh * p.Sq + qrow] = (unsigned char)QATTN_PATH_TWO_TERM;   // (one-term sweeps leave the call's pre-fill)
    }
    return 0;
}

// Everything a wave derives from its thread / block index for one pass over its query rows, and that pass itself (the Q^T
// fragments are re-loaded by a second pass: a few KiB against the pass's megabytes of K / V).
// SUMM (MEASURED AND NOT USED, round 4: every instantiation passes false): the row sums of the QUANTISED hi and lo terms from the matrix
// pipe -- two v_mfma_f32_16x16x128 per chunk on the operands the PV products consume (full_step) -- instead of 36 fp32 additions per chunk
// and lane.  The two-term sweep is bound by vector-instruction issue, and this took 4.8 % off it (profiles/r04/ab_two_term_valu_diet.log) --
// but it breaks the bound on very peaked rows (q x 3 at S = 4096: 0.022): thousands of keys 15 binades below the row's top key flush to
// zero in fp8; their V rows average out of the numerator, but their weights ARE 0.5 .. 1 % of the denominator, and a denominator that
// drops them too rescales the output by that much.  The exact fp32 sum of the un-rounded exponentials stays.
template <int D, int NW, int QK_FMT, int V_FMT, bool CAUSAL, bool TOKEN, bool TWO, bool BYTE, int ABL, bool Q16, bool NEFF = false, bool SUMM = false, int IN16 = QATTN_FMT_BF16>
__device__ __forceinline__ int block_pass(const AttnParams& p, unsigned char* smem, int tid, int bid, bool check_peaked, volatile unsigned* mail = nullptr) {
    constexpr int CH = 64 * D;      // bytes of one K (or V) chunk
    constexpr int STAGE = 2 * CH;   // K chunk + V chunk
    constexpr int KS = D / 64;      // QK^T k-steps
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hh = lane >> 5;

    int head, qb;
    map_block(p, bid, p.nqb, CAUSAL, head, qb);
    const int b = head / p.Hq, h = head % p.Hq;
    const int hkv = h / (p.Hq / p.Hkv);
    const long bh = (long)b * p.Hq + h;
    const long kv_head = (long)b * p.Hkv + hkv;
    constexpr int QWG = NW * kQPerWave;
    const int q0_wg = qb * QWG;
    const int q0 = q0_wg + wave * kQPerWave;  // first query row of this wave
    const int qrow = q0 + ql;                 // this lane's query row

    const unsigned char* kg = p.k + kv_head * (long)p.nchunks * CH;
    const unsigned char* vg = p.v + kv_head * (long)p.nchunks * CH;

    // chunks the workgroup / this wave must visit (causal: up to the diagonal of the last row)
    const int n_wg = CAUSAL ? min(p.nchunks, (min(q0_wg + QWG, p.Sq) - 1) / 64 + 1) : p.nchunks;
    const int n_w = CAUSAL ? min(n_wg, (q0 + kQPerWave - 1) / 64 + 1) : p.nchunks;

    // Q^T fragments: global -> this lane's own slots of the workgroup's Q area in LDS (behind the K/V ring);
    // only the writing lane ever reads them back, so no barrier is needed (the compiler orders the lane's own
    // ds_write -> ds_read with lgkmcnt).  NW words behind the Q area collect the waves' "a row of mine is peaked" votes.
    unsigned char* qbuf = smem + kStagesV2 * STAGE + wave * (KS << 11) + (hh << 10) + (ql << 4);
    unsigned* vote = reinterpret_cast<unsigned*>(smem + v2_words_offset<D, NW, Q16>());
    // fused step: the scale bytes of this head's V chunks (block-scaled V; 127 = 2^0 where V has one scale per head), kept in LDS
    // behind the votes for the PV products of every pass of this block; the sweep's first barrier publishes them
    unsigned* vx = vote + 16;
    // fused step: everything the head of a block needs from memory is requested at once -- the V scale byte of this thread, the
    // abs-max words, then the wave's 16-bit Q rows -- and waited for once.  Left where they are used (the Q rows inside the
    // sweep's prologue) the compiler waited for each small load in turn and only then asked for Q: four memory round trips in
    // a row at the head of every block.  (At this point the previous block's O stores are still in flight, so any wait is a
    // vmcnt(0): one wait covers all of them, whatever their order.)
    uint4 rawq[Q16 ? KS : 1][4];
    float scale_q16 = 1.0f;
    constexpr bool kStagesFirst = NW == 8;
    if (kStagesFirst) {   // stage 0 = {K(0), V(0)}, stage 1 = {K(1), V(0)} (kv_sweep's dma_next, first two calls)
        const unsigned koff1 = min((unsigned)CH, (unsigned)(p.nchunks - 1) * CH);
        stage_dma8<D>(kg, vg, 0u, 0u, smem, 0u, wave, lane);
        stage_dma8<D>(kg, vg, koff1, 0u, smem, (unsigned)STAGE, wave, lane);
    }
    if (Q16) {
        static_assert(!Q16 || !TOKEN, "the fused Q path is head-wise");
        // (branch-free, clamped indices: a conditional load or a loop with a run-time trip count gets its own wait)
        const unsigned* part = p.q_amax_part + bh * p.amax_stride;
        const unsigned* vrow = p.vexp ? p.vexp + kv_head * p.vexp_stride : part;   // (vexp only when nchunks <= kVxWords; else any readable word)
        const int vmax = p.vexp ? p.nchunks - 1 : 0, amax_last = p.amax_n - 1;
        static_assert(kMomentSplits <= 256, "four abs-max words per lane");
        unsigned vxw = vrow[min(tid & (kVxWords - 1), vmax)];
        const unsigned a0 = part[min(lane, amax_last)], a1 = part[min(lane + 64, amax_last)];
        const unsigned a2 = part[min(lane + 128, amax_last)], a3 = part[min(lane + 192, amax_last)];
        const bool qvalid = qrow < p.Sq;
        const uint4* qp = reinterpret_cast<const uint4*>(q16_row(p, b, h, bh, qvalid ? qrow : 0, D * 2) + hh * 64);
#pragma unroll
        for (int s = 0; s < KS; s++)
#pragma unroll
            for (int i = 0; i < 4; i++) rawq[s][i] = qp[s * 8 + i];   // 8 elements each; the lane's 32 elements d = 64s + 32hh .. +31
        asm volatile("" ::: "memory");   // (keeps the requests above what follows)
        static_assert(kVxWords <= NW * 64, "one V scale word per thread");
        if (!(p.vexp && tid < p.nchunks)) vxw = 127u;
        if (tid < kVxWords) vx[tid] = (unsigned)vscale_word(vxw);
        const unsigned am = max(max(a0, a1), max(a2, a3)) & 0x7fffffffu;   // (a caller-supplied abs-max enters by magnitude)
        const float inv_qmax = (float)(1.0 / (double)(QK_FMT == QATTN_FMT_E4M3 ? 448.0 : 57344.0));
        scale_q16 = make_scale(__uint_as_float(wave_allmax_u32(am)), inv_qmax, p.q_numerics, IN16);
        if (q0_wg == 0 && tid == 0) p.sq_out[bh] = scale_q16;
    }
    // pre-quantised Q: the lane's 32-byte pieces, requested here for the same reason
    v4i rawq8[Q16 ? 1 : KS][2];
    if (!Q16) {
        const bool qvalid = qrow < p.Sq;
        const unsigned char* qp = p.q + ((bh * p.Sq + (qvalid ? qrow : 0)) * D) + hh * 32;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            rawq8[s][0] = *reinterpret_cast<const v4i*>(qp + s * 64);
            rawq8[s][1] = *reinterpret_cast<const v4i*>(qp + s * 64 + 16);
        }
        asm volatile("" ::: "memory");
    }
    // softmax scale in the exp2 domain: c = scale_q * scale_k * sm_scale * log2(e)   (tk/attention.py:204-210); the per-head
    // scales through the scalar cache (a vector load of a uniform value is waited for with vmcnt(0): every load above)
    float c;
    if (TOKEN) c = p.sm_log2e * (qrow < p.Sq ? p.sq[bh * p.Sq + qrow] : 1.0f);
    else if (Q16) c = p.sm_log2e * scale_q16 * scalar_load_f32(p.sk + kv_head);
    else c = p.sm_log2e * scalar_load_f32(p.sq + bh) * scalar_load_f32(p.sk + kv_head);
    const float* skt = TOKEN ? p.sk + kv_head * p.Skv : nullptr;

    // invoked by kv_sweep right after the first K/V stages have been requested, so the two latencies overlap
    auto load_q = [&]() {
        const bool qvalid = qrow < p.Sq;
        if (Q16) {
            const float rinv = 1.0f / scale_q16;
#pragma unroll
            for (int s = 0; s < KS; s++) {
                int2 w[4];
#pragma unroll
                for (int i = 0; i < 4; i++) w[i] = quant8<IN16, QK_FMT>(qvalid ? rawq[Q16 ? s : 0][i] : make_uint4(0, 0, 0, 0), scale_q16, rinv);
                *reinterpret_cast<v4i*>(qbuf + (s << 11)) = v4i{w[0].x, w[0].y, w[1].x, w[1].y};
                *reinterpret_cast<v4i*>(qbuf + (s << 11) + 512) = v4i{w[2].x, w[2].y, w[3].x, w[3].y};
            }
        } else {
#pragma unroll
            for (int s = 0; s < KS; s++) {
                v4i lo = rawq8[Q16 ? 0 : s][0], hi = rawq8[Q16 ? 0 : s][1];
                if (!qvalid) { lo = v4i{0, 0, 0, 0}; hi = v4i{0, 0, 0, 0}; }
                *reinterpret_cast<v4i*>(qbuf + (s << 11)) = lo;
                *reinterpret_cast<v4i*>(qbuf + (s << 11) + 512) = hi;
            }
        }
    };
    // head-wise one-term byte-exponential kernels have 16 registers to spare and hold the Q^T fragments in them
    constexpr bool QREG = BYTE && !TWO && !TOKEN;
    // (Measured and removed: prefetching the NEXT block's Q rows near the end of a static non-causal sweep -- LDS-DMA into a dump slot, to have
    // them in this XCD's L2 when the next prologue asks: fused step auto +0.3 %, fast 0 %, non-temporal form +0.5 %; 64 KiB of once-read
    // bf16 rows per block and CU compete with the two live heads' K / V for the XCD's 4 MiB L2.  profiles/r04/ab_c2_q_prefetch_variants.log;
    // the code is in the history up to round 5.)
    return attend_block<D, NW, QK_FMT, V_FMT, CAUSAL, TOKEN, TWO, BYTE, ABL, QREG, Q16, NEFF && !TWO, SUMM && TWO && !BYTE>(
        p, smem, kg, vg, qbuf, vote, n_wg, n_w, q0, qrow, wave, lane, bh, kv_head, c, skt, check_peaked, load_q, vx, kStagesFirst, mail,
        out_head_offset(p, b, h));
}

// The rescue of a block's flagged 32-row groups as a pass of its own, run by whichever workgroup took the queue item (or by the
// block's own workgroup in a static launch): everything is derived afresh from the (opaque) thread index and the block id, so
// that nothing of a sweep is live here and nothing of this is live in a sweep.  The rescued rows' Q^T fragments are fetched
// from global memory (fused step: the 16-bit rows, quantised with the quant8 sequence of the sweep's prologue: the same
// bytes), the head's V chunk scales are re-read into LDS; the K/V ring is idle and holds the prefetch and the merge.
// The e-th set bit of m (e < popcount(m)): five halving steps on prefix popcounts.
__device__ __forceinline__ int select_bit(unsigned m, int e) {
    int pos = 0;
#pragma unroll
    for (int step = 16; step >= 1; step >>= 1)
        if (__builtin_popcount(m & ((1u << (pos + step)) - 1u)) <= e) pos += step;
    return pos;
}

template <int D, int NW, int QK_FMT, int V_FMT, bool CAUSAL, bool Q16, int IN16 = QATTN_FMT_BF16>
__device__ __forceinline__ void rescue_pass(const AttnParams& p, unsigned char* smem, int tid, int bid, int nrows_flag) {   // set bits of the block's vote words (| kRescueSevere)
    const int nrows = nrows_flag & (kRescueSevere - 1);
    const bool on16 = Q16 && (nrows_flag & kRescueSevere) != 0;   // (fused step) the 16-bit-V rescue, workgroup-uniform
    constexpr int CH = 64 * D;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hh = lane >> 5;
    int head, qb;
    map_block(p, bid, p.nqb, CAUSAL, head, qb);
    const int b = head / p.Hq, h = head % p.Hq;
    const long bh = (long)b * p.Hq + h;
    const long kv_head = (long)b * p.Hkv + h / (p.Hq / p.Hkv);
    const unsigned char* kg = p.k + kv_head * (long)p.nchunks * CH;
    const unsigned char* vg = p.v + kv_head * (long)p.nchunks * CH;
    // (fused step, two-term rescue on the block-scaled fp8 V: the head's chunk scale words, left in LDS by the block's sweep -- nothing
    // between the sweep and this pass touches the words' corner of LDS)
    const unsigned* vx = reinterpret_cast<const unsigned*>(smem + v2_words_offset<D, NW, Q16>()) + 16;
    float c, scale_q16 = 1.0f;
    if (Q16) {
        const float inv_qmax = (float)(1.0 / (double)(QK_FMT == QATTN_FMT_E4M3 ? 448.0 : 57344.0));
        scale_q16 = make_scale(__uint_as_float(max_partials(p.q_amax_part + bh * p.amax_stride, p.amax_n, lane) & 0x7fffffffu), inv_qmax, p.q_numerics, IN16);
        c = p.sm_log2e * scale_q16 * p.sk[kv_head];
    } else {
        c = p.sm_log2e * p.sq[bh] * p.sk[kv_head];
    }
    // the block's peaked rows, wave after wave in row order, 32 per group: lane pair ql takes entry 32 g + ql (a group's spare lanes
    // recompute its first row and store nothing)
    unsigned masks[NW];
    {
        const volatile unsigned* vote = reinterpret_cast<const volatile unsigned*>(smem + v2_words_offset<D, NW, Q16>());
        v4i va, vb;
        lds_read_8words_raw(vote, va, vb);
#pragma unroll
        for (int w = 0; w < NW; w++) masks[w] = (unsigned)(w < 4 ? va[w & 3] : vb[w & 3]);
    }
    if (on16) lds_barrier();   // (the 16-bit-V rescue's V areas cover the vote words: every wave has read them before any area is filled)
    for (int g0 = 0; g0 < nrows; g0 += kQPerWave) {
        const bool have = g0 + ql < nrows;
        int e = have ? g0 + ql : g0, wsel = 0;
        unsigned msel = masks[0];
#pragma unroll
        for (int w = 0; w < NW - 1; w++) {   // walk to the wave that holds entry e (branch-free: at most NW - 1 steps)
            const int cnt = __builtin_popcount(masks[w]);
            const bool next = wsel == w && e >= cnt;
            e = next ? e - cnt : e;
            msel = next ? masks[w + 1] : msel;
            wsel = next ? w + 1 : wsel;
        }
        const int row = qb * (NW * kQPerWave) + wsel * kQPerWave + select_bit(msel, e);
        // (wave-uniform bounds of the group's rows: its first entry is its lowest row, its last valid entry the highest)
        const int row_lo = __builtin_amdgcn_readfirstlane(row);
        const int row_hi = __builtin_amdgcn_readlane(row, min(nrows - g0, kQPerWave) - 1);
        const bool qvalid = row < p.Sq;
        auto qfrag = [&](int s_) -> v8i {
            if (Q16) {
                const float rinv = 1.0f / scale_q16;
                const uint4* qp = reinterpret_cast<const uint4*>(q16_row(p, b, h, bh, qvalid ? row : 0, D * 2) + hh * 64) + s_ * 8;
                int2 w[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    uint4 raw = qp[i];
                    if (!qvalid) raw = make_uint4(0, 0, 0, 0);
                    w[i] = quant8<IN16, QK_FMT>(raw, scale_q16, rinv);
                }
                return v8i{w[0].x, w[0].y, w[1].x, w[1].y, w[2].x, w[2].y, w[3].x, w[3].y};
            } else {
                const unsigned char* qp = p.q + ((bh * p.Sq + (qvalid ? row : 0)) * D) + hh * 32 + s_ * 64;
                v4i lo = *reinterpret_cast<const v4i*>(qp), hi = *reinterpret_cast<const v4i*>(qp + 16);
                if (!qvalid) { lo = v4i{0, 0, 0, 0}; hi = v4i{0, 0, 0, 0}; }
                return v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        };
        // parked in this lane's own slots of the workgroup's Q area (behind the ring), as the sweep does: held in registers for the
        // rescue loop they cost the causal fused instantiation two spilled registers
        // (fused step: behind the eight V areas of the 16-bit-V rescue, qattn_pv16.h rescue_rows16_at)
        unsigned char* qslot = smem + (on16 ? kRescue16VBytes : kStagesV2 * 2 * 64 * D) + wave * ((D / 64) << 11) + (hh << 10) + (ql << 4);
#pragma unroll
        for (int s_ = 0; s_ < D / 64; s_++) {
            const v8i f = qfrag(s_);
            *reinterpret_cast<v4i*>(qslot + (s_ << 11)) = v4i{f[0], f[1], f[2], f[3]};
            *reinterpret_cast<v4i*>(qslot + (s_ << 11) + 512) = v4i{f[4], f[5], f[6], f[7]};
        }
        bool done16 = false;
        if constexpr (Q16) {
            if (on16) {
                // the fused step has the original 16-bit V at hand: where a flagged row's weight sits on very few keys (kPeakR16) -- its output
                // carries V's rounding nearly one to one -- the group gets the reference kernel's own P.V numerics (16-bit P, 16-bit V)
                const unsigned char* vg16 = v16_head(p, b, h / (p.Hq / p.Hkv), kv_head, D * 2);
                rescue_rows16_at<D, NW, QK_FMT, IN16, CAUSAL>(p, smem, kg, vg16, row, have, row_lo, row_hi, wave, lane, bh, c,
                                                              [&](int s_) { return lds_read_frag(qslot + (s_ << 11)); });
                done16 = true;
            }
        }
        if (!done16) {
            rescue_rows_at<D, NW, QK_FMT, V_FMT, CAUSAL, false, true, true>(p, smem, kg, vg, row, have, row_lo, row_hi, wave, lane, bh, kv_head, c, nullptr,
                                                                           [&](int s_) { return lds_read_frag(qslot + (s_ << 11)); }, Q16 ? vx : nullptr);
            if (p.path && wave == 0 && hh == 0 && have && qvalid) p.path[bh * p.Sq + row] = (unsigned char)QATTN_PATH_TWO_TERM;   // (fused entry's debug output)
        }
    }
}

// One 256-row block: a block whose rows are predicted peaked (predicted_r) starts two-term; a one-term pass that finds too
// many peaked rows loops back into the same two-term code.
// Returns the bit mask of the block's 32-row groups (waves) that still need rescue_pass (0: none; CHECK launches only).
template <int D, int NW, int QK_FMT, int V_FMT, bool CAUSAL, bool TOKEN, bool BYTE, int ABL, bool Q16, bool CHECK, int IN16 = QATTN_FMT_BF16>
__device__ __forceinline__ unsigned run_block(const AttnParams& p, unsigned char* smem, int bid, int tid, volatile unsigned* mail) {   // tid: an opaque copy of threadIdx.x; mail: draw_next_block
    int head, qb;
    map_block(p, bid, p.nqb, CAUSAL, head, qb);
    // Fused step (the original bf16 V is at hand): a block whose first row sees fewer than two_term_keys keys -- early causal rows, short
    // sequences -- runs the reference kernel's own P.V numerics, 16-bit P on the un-quantised V (qattn_pv16.h), instead of two-term fp8
    // P on the fp8 V: row 0 of a causal head IS V[0], and an fp8 V puts its rounding error (up to 2^-4 relative) straight into the output.
    // The same MFMA time as the two-term pass (16 bf16 products of 32 cycles for 8 fp8 products of 64), fewer VALU instructions.
    // ... and so does (round 5) every block of the fused step that used to run two-term fp8 P: blocks predicted peaked, blocks that
    // repeat after a one-term sweep with too many flagged rows, every block under QATTN_PRECISION_ACCURATE (`pass16` below).  The same
    // MFMA time as the two-term pass, fewer vector instructions, and the rows that need the precision most no longer attend an fp8 V.
    auto pass16 = [&](volatile unsigned* mail_) {
        asm volatile("" : "+v"(tid));
#ifndef QATTN_PV16_PIPELINED
#define QATTN_PV16_PIPELINED 1   // (a build knob for tools/ab.py variants: 0 = the un-pipelined pass of round 4 everywhere)
#endif
        if constexpr (NW == 8) {
            if constexpr (QATTN_PV16_PIPELINED != 0)
                pv16p_block_pass<D, NW, QK_FMT, IN16, CAUSAL>(p, smem, tid, bid, [&]() { return draw_issue(p, mail_, tid); },
                                                              [&](unsigned ticket) { draw_finish(p, mail_, tid, ticket); });
            else
                pv16_block_pass<D, NW, QK_FMT, IN16, CAUSAL, false, true>(
                    p, smem, tid, bid, [&]() { return draw_issue(p, mail_, tid); }, [&](unsigned ticket) { draw_finish(p, mail_, tid, ticket); });
        }
    };
    constexpr bool kPass16 = Q16 && NW == 8;   // (the fused step always carries the 16-bit V: qattn_api.hip quant_attention_impl)
    if constexpr (kPass16) {
        const int nkeys0 = CAUSAL ? min(p.Skv, qb * (NW * kQPerWave) + 1) : p.Skv;
        if (nkeys0 < p.two_term_keys) {   // workgroup-uniform
            pass16(mail);
            return 0u;
        }
    }
    // One copy of each pass.  Every per-lane value is re-derived inside block_pass from an opaque copy of the thread index, so
    // nothing of the one-term pass stays live in registers across the two-term loop (and vice versa).
    bool two = p.n_two != 0;  // workgroup-uniform; n_two = nqb: every block (QATTN_PRECISION_ACCURATE)
    if (!two) {
        float var = 1.0f;
        if (p.ssq_q) {
            // (lane from the caller's opaque thread index: derived from threadIdx.x it is hoisted out of the kernel's block loop
            // as a 64-bit byte offset and spilled)
            const int lane = tid & 63, kvh = (head / p.Hq) * p.Hkv + (head % p.Hq) / (p.Hq / p.Hkv);
            float sa, sb;
            sum_partials_pair(p.ssq_q + (long)head * p.ssq_stride, p.ssq_k + (long)kvh * p.ssq_stride, p.ssq_n, lane, sa, sb);
            var = sa * sb * p.var_mul;
            if (!(var >= kVarDeadband)) var = 1.0f;
        }
        // (Round 5, measured and dropped: the sums REQUESTED here and looked at inside the one-term pass once its own small words had returned
        // -- no round trip of their own in front of the block, 1.0 us of 64 in the dev work log -- with a block that then belongs on the other
        // pass leaving the one-term pass at once: bit-identical, C2 -0.1 %, C3 -0.9 %, C5 -0.5 %, but q x 2 +1.4 % for the abandoned first
        // requests; profiles/r05/ab_ssq_prefetch_vs_head_dropped.log.  With a per-workgroup hint word that takes the test up front again after a
        // "wide" verdict the q x 2 cost goes, and so does most of the gain: ab_ssq_hint_and_light_last_vs_head_dropped.log.)
        const int nkeys = CAUSAL ? min(p.Skv, qb * (NW * kQPerWave) + 1) : p.Skv;   // keys the block's first row attends
        // unit variance (or no estimate): the key-count rule; a head that IS wide: two-term when so many rows are expected to end peaked
        // that gathering and recomputing them would cost more than the two-term sweep
        const bool wide = var >= kVarDeadband;
        // (the key-count rule as the integer comparison it stands for: predicted_r(nkeys, 1, peak_z) < kPeakR0 <=> nkeys < two_term_keys
        // up to float rounding -- which at nkeys == two_term_keys EXACTLY, a non-causal call with 1024 keys, fell on the wrong side and
        // started every block on the precise pass, FAST included: found by the strict per-row-path grader, tools/fuzz_parity.py seed 182 #191)
        const bool start_two = nkeys < p.two_term_keys || (wide && many_rows_peaked((float)nkeys, var));
        two = __builtin_amdgcn_readfirstlane(start_two ? 1 : 0) != 0;   // (every lane holds the same value)
    }
    unsigned to_rescue = 0u;
    for (;;) {
        asm volatile("" : "+v"(tid));
        if (two) {
            if constexpr (kPass16) pass16(mail);
            else block_pass<D, NW, QK_FMT, V_FMT, CAUSAL, TOKEN, true, false, 0, Q16, false, false, IN16>(p, smem, tid, bid, false, mail);   // (SUMM = false: see block_pass)
            break;
        }
        const int r = block_pass<D, NW, QK_FMT, V_FMT, CAUSAL, TOKEN, false, BYTE, ABL, Q16, CHECK, false, IN16>(p, smem, tid, bid, CHECK, mail);
        mail = nullptr;   // (a repeated block has drawn its successor already)
        if (r == 0) break;
        if (r != kPassRedo) {   // a few peaked groups (every wave is past the vote barrier, hence done with the K/V ring)
            to_rescue = (unsigned)r;
            break;
        }
        two = true;  // many rows of this block are peaked: the block repeats in two-term mode
    }
    return to_rescue;
}

// QK_FMT / V_FMT: QATTN_FMT_E4M3 (0) or QATTN_FMT_E5M2 (1) == the MFMA's cbsz/blgp selector.
// BYTE: the one-term pass uses the byte exponential (else exact v_exp_f32 + RNE conversion, needed for the LSE output).
// Q16: the fused step (qattn_fp8_quant_attention_forward): Q arrives as bf16 and is quantised here, row by row, with the
// same quant8 sequence as the pre-pass (bit-identical q8), from the head's abs-max bits -- the pre-pass then neither
// re-reads Q nor writes q8, and this kernel reads 2 instead of 1 byte per Q element once.
// CHECK: QATTN_PRECISION_AUTO -- one-term passes carry the two peakedness statistics (R and the effective key count) and end
// with the vote / rescue / redo logic; the FAST launches instantiate neither.
// One launch covers every query block of every head, and launches are PERSISTENT: one workgroup per CU (LDS allows no more),
// no workgroup launch, LDS allocation and wave start between blocks (-1.6 % fast, -2.5 % auto at C2 against one workgroup
// per block).  Non-causal: blocks are equal, the workgroups walk them with a fixed stride (blockIdx.x, + gridDim.x, ...: the
// blocks the hardware would have handed that XCD one by one, bid & 7 is preserved).  Causal: blocks differ 16 : 1, so with a
// scheduler state in the workspace (SchedState, qattn_attn.h) a workgroup starts on block blockIdx.x and draws its next blocks
// from its XCD label's counter -- the order in which the hardware would have handed that XCD its workgroups: a head's K/V stay
// in that XCD's L2 and the order is longest-processing-time first; a workgroup whose counter has run dry takes blocks of the
// other XCDs (S = 8192: -4 %, S = 16384: -2...3 % against one workgroup per block; S = 4096: equal).  A static stride over that
// order was 6 % slower and balanced pairs of blocks 3-10 % slower in round 2.  Without the state (a caller without a
// workspace) causal launches use one workgroup per block.
// SV: kStrided16 of the translation unit, part of the kernel's NAME only -- the dense (qattn_attn_v2_*) and the strided-view (qattn_attn_v2_*_sv)
// units instantiate the same template arguments with different bodies (qattn_attn.h QATTN_STRIDED16).
template <int D, int NW, int QK_FMT, int V_FMT, bool CAUSAL, bool TOKEN, bool BYTE, int ABL = 0, bool Q16 = false, bool CHECK = false, int IN16 = QATTN_FMT_BF16,
          int SV = QATTN_STRIDED16>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_kernel_v2(const AttnParams p_arg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // two words behind the waves' vote words carry a block number / queue item from thread 0 to the workgroup
    volatile unsigned* bcast = reinterpret_cast<volatile unsigned*>(smem + v2_words_offset<D, NW, Q16>()) + 8;
#if defined(__HIP_DEVICE_COMPILE__)
    // the parameters are re-read from the kernel-argument segment every block (scalar loads): left to the compiler they
    // are hoisted out of the block loop, stay live across whole blocks and push the scalar file into spilling
    typedef const __attribute__((address_space(4))) AttnParams* KernargPtr;
#define QATTN_PARAMS()                                                                                   \
    KernargPtr pk_ = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr(); /* AttnParams is the only argument */ \
    asm volatile("" : "+s"(pk_));                                                                        \
    const AttnParams& p = *(const AttnParams*)pk_
#else
#define QATTN_PARAMS() const AttnParams& p = p_arg
#endif
    const bool dynamic = p_arg.sched != nullptr;   // launch-uniform
    // The thread index is rebuilt every round from the wave's number (a scalar) and v_mbcnt (volatile: not hoisted): threadIdx.x
    // kept live across the block loop costs a vector register the tightest instantiations do not have (one spilled dword).
    const int wave_s = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    int bid = blockIdx.x;
    int parity = 0;   // the mailbox alternates between two words: a wave still reading this block's successor cannot meet the next draw
    for (;;) {
        int tid;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(tid));
        tid |= wave_s << 6;
        asm volatile("" : "+v"(tid));
        int resc;
        {
            QATTN_PARAMS();
            resc = run_block<D, NW, QK_FMT, V_FMT, CAUSAL, TOKEN, BYTE, ABL, Q16, CHECK, IN16>(p, smem, bid, tid, dynamic ? bcast + parity : nullptr);
        }
        int next_drawn = -1;
        if constexpr (CHECK && !TOKEN && NW == 8 && Q16) {
            // (the 16-bit-V rescue's V areas cover the mailbox: the successor's number -- written before the block's vote barrier -- is taken out first)
            if ((resc & kRescueSevere) != 0 && dynamic) next_drawn = __builtin_amdgcn_readfirstlane((int)lds_read_word_raw(bcast + parity));
        }
        if constexpr (CHECK && !TOKEN && NW == 8) {
            // a few peaked rows: on the spot, while the head's K / V are in this XCD's L2 (every wave is past the vote barrier, hence
            // done with the K/V ring).  The block's successor HAS been drawn by then (draw_issue / draw_finish inside attend_block, before the
            // vote barrier: requested behind the row stores the atomic cost every block 1.2 - 2 us); it waits for the rescue -- the
            // trade round 4 measured (C3 -2.8 % against drawing after the stores; a block held through a rescue adds to the tail).
            if (resc != 0) {
                QATTN_PARAMS();
                asm volatile("" : "+v"(tid));
                rescue_pass<D, NW, QK_FMT, V_FMT, CAUSAL, Q16, IN16>(p, smem, tid, bid, resc);
            }
        }
        QATTN_PARAMS();
        int next = -1;
        if (dynamic) {
            // the successor was drawn in this block's prologue (draw_next_block)
            lds_barrier();   // also: every wave has left the ring and the Q slots before the next block fills them
            next = (CHECK && !TOKEN && NW == 8 && Q16 && (resc & kRescueSevere) != 0) ? next_drawn : __builtin_amdgcn_readfirstlane((int)lds_read_word_raw(bcast + parity));
            parity ^= 1;
        } else if (!CAUSAL && bid + (int)gridDim.x < p.total_blocks) {
            next = bid + (int)gridDim.x;
            lds_barrier();
        }
        if (next < 0) break;
        bid = next;
    }
#undef QATTN_PARAMS
}

template <int D, int NW, int FMT, bool CAUSAL, bool TOKEN, bool BYTE, bool Q16, bool CHECK, int IN16 = QATTN_FMT_BF16>
static int launch_attn_v2_chk(const AttnParams& pin, hipStream_t st) {
    AttnParams p = pin;
    p.total_blocks = p.B * p.Hq * p.nqb;
    // one persistent workgroup per CU (a multiple of 8 keeps every workgroup's blocks on one XCD: blocks b and b + grid share
    // b & 7; xcd_remap is only set on an 8-XCD device, qattn_api.hip); fewer blocks than CUs: one each
    const int cus = p.xcd_remap ? cu_count() & ~7 : cu_count();
    const bool persistent = p.persistent && cus >= 8 && p.total_blocks > cus;
    int grid = persistent && (!CAUSAL || p.sched) ? cus : p.total_blocks;
    if (p.sched && persistent && (CAUSAL || p.total_blocks >= (long)p.dyn_min_rounds * cus)) {
        // dynamic hand-out: per-XCD-label counters, zeroed every call (by a one-block kernel: qattn_attn.h zero_words).  Causal launches
        // always (unequal blocks); non-causal ones when a workgroup has many rounds to go: the CUs of one chip differ by +-5 %
        // in speed under this kernel (dev work log: finish times spread over 58 us of a 597 us C2 launch), which equal shares
        // turn into idle time at the end.  Measured (tools/ab.py, profiles/r03/ab_dyn_noncausal.log): 32 rounds -1.6 %,
        // 16 rounds -0.3 %, 8 rounds +0..2.5 % (a block is then too coarse a unit to even anything out).
        p.sched_nq = p.xcd_remap ? 8 : 1;
        if (!p.sched_zeroed && zero_words(p.sched->next, (long)(sizeof(SchedState) / sizeof(unsigned)), st) != hipSuccess) return QATTN_ERR_LAUNCH;
    } else {
        p.sched = nullptr;
    }
    size_t lds = (size_t)kStagesV2 * 2 * 64 * D + (size_t)NW * kQPerWave * D + 64 + 4 * kVxWords + 1024;  // K/V ring + parked Q^T fragments + per-wave vote words + V chunk scale bytes + 1 KiB spare (the removed Q prefetch's dump slot; kept: the allocation is part of the measured launch)
    if (Q16 && NW == 8) lds = (size_t)kLdsAll;   // the fused kernels: the pipelined 16-bit-V pass's ring (5 x 24 KiB), the 16-bit-V rescue's V areas + parked Q^T fragments (all 160 KiB), the words at the end (v2_words_offset)
    static_assert(kP16Stages * (64 * 128 + 64 * 2 * 128) <= kLdsAll - 4096 && kStagesV2 * 2 * 64 * 128 + 8 * kQPerWave * 128 <= kLdsAll - 4096, "the rings stay clear of the words");
    //  the fused kernels' words -- 16 vote / mailbox words, kVxWords V scale words -- live in the last 4 KiB, which
    // the 16-bit-V rescue's parked Q^T fragments (wave 7's slot, behind its eight V areas) deliberately cover: rescue_pass takes the masks and
    // the kernel's block loop the drawn successor OUT before anything is parked there (read-before-clobber, see their comments)
    static_assert(kRescue16VBytes + 8 * kQPerWave * 128 == kLdsAll, "the 16-bit-V rescue's V areas + parked Q^T fragments fill the CU's LDS exactly");
    static_assert(64 + 4 * kVxWords <= 4096, "vote words + V scale words fit the last 4 KiB");
    if constexpr (FMT == QATTN_FMT_E4M3 && Q16 && BYTE && NW == 8 && !TOKEN && IN16 == QATTN_FMT_BF16 && !kStrided16) {
        if (p.stamp_buf) {   // measurement entry: the same kernel with the two clock stamps per wave
            auto kern1 = attn_fwd_kernel_v2<D, NW, FMT, FMT, CAUSAL, TOKEN, BYTE, 1024, Q16, CHECK>;
            if (hipFuncSetAttribute((const void*)kern1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return QATTN_ERR_LAUNCH;
            hipLaunchKernelGGL(kern1, dim3(grid), dim3(NW * 64), lds, st, p);
            return QATTN_OK;
        }
    }
    if (p.stamp_buf) return QATTN_ERR_UNSUPPORTED_FMT;   // (only the fused e4m3 step has a stamped instantiation)
    auto kern = attn_fwd_kernel_v2<D, NW, FMT, FMT, CAUSAL, TOKEN, BYTE, 0, Q16, CHECK, IN16>;
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return QATTN_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, p);
    return QATTN_OK;
}
template <int D, int NW, int FMT, bool CAUSAL, bool TOKEN, bool BYTE, bool Q16 = false, int IN16 = QATTN_FMT_BF16>
static int launch_attn_v2_one(const AttnParams& p, hipStream_t st) {
    return p.peak_r0 > 0.0f ? launch_attn_v2_chk<D, NW, FMT, CAUSAL, TOKEN, BYTE, Q16, true, IN16>(p, st)
                            : launch_attn_v2_chk<D, NW, FMT, CAUSAL, TOKEN, BYTE, Q16, false, IN16>(p, st);
}

// The fused step from fp16 inputs (round 5): the same kernels with IN16 = QATTN_FMT_FP16 -- in-kernel Q quantisation (quant8's fp16 fast
// path), block-scaled V, 16-bit-V passes on fp16 V and P.  Instantiated in translation units of their own (build.py: -DQATTN_ONLY_IN16=3
// beside -DQATTN_ONLY_FMT; the other units compile with QATTN_ONLY_IN16=2 and only call the entry points).
// Entry points of the strided-view units carry the suffix _sv (build.py: -DQATTN_V2_SV; the dense units: -DQATTN_STRIDED16=0).
#ifdef QATTN_V2_SV
#define QATTN_V2_ENTRY(name) name##_sv
static_assert(kStrided16, "a strided-view unit addresses through the strides");
#else
#define QATTN_V2_ENTRY(name) name
static_assert(!kStrided16, "the dense units of the hand-scheduled kernel are built with -DQATTN_STRIDED16=0 (qattn_attn.h)");
#endif
int QATTN_V2_ENTRY(launch_attn_v2_f16_e4m3)(const AttnParams& p, int causal, hipStream_t st);
int QATTN_V2_ENTRY(launch_attn_v2_f16_e5m2)(const AttnParams& p, int causal, hipStream_t st);
template <int FMT, bool CAUSAL>
static int launch_attn_v2_f16(const AttnParams& p, hipStream_t st) {
    return FMT == QATTN_FMT_E4M3 ? QATTN_V2_ENTRY(launch_attn_v2_f16_e4m3)(p, CAUSAL ? 1 : 0, st) : QATTN_V2_ENTRY(launch_attn_v2_f16_e5m2)(p, CAUSAL ? 1 : 0, st);
}

template <int D, int NW, int FMT, bool CAUSAL>
static int launch_attn_v2_t(const AttnParams& pin, int scale_mode, hipStream_t st) {
    AttnParams p = pin;
    // blocks that run two-term P from the start: every block (QATTN_PRECISION_ACCURATE), else the kernel's predicted_r rule
    // (at unit score variance: the blocks whose first row sees fewer than kTwoTermKeys keys)
    p.n_two = p.precision == QATTN_PRECISION_ACCURATE ? p.nqb : 0;
    // byte-exponential one-term pass, unless the caller wants the LSE (needs the exact row sum) or exact exponentials
    const bool byte_exp = p.lse == nullptr;
    if (scale_mode == QATTN_SCALE_TOKEN) return QATTN_ERR_UNSUPPORTED_FMT;  // routed to the templated kernel (attn_v2_covers)
    if (p.q16 != nullptr) {   // fused step: byte path only (qattn_api.hip); its 16-bit input type is the output's
        if (p.out_fmt == QATTN_FMT_FP16) return launch_attn_v2_f16<FMT, CAUSAL>(p, st);   // (a translation unit of its own, below)
        return launch_attn_v2_one<D, 8, FMT, CAUSAL, false, true, true>(p, st);
    }
#ifdef QATTN_V2_SV
    (void)byte_exp;
    return QATTN_ERR_UNSUPPORTED_FMT;   // (the strided-view units hold the fused step's kernels only: nothing else reads a 16-bit tensor)
#else
    return byte_exp ? launch_attn_v2_one<D, NW, FMT, CAUSAL, false, true>(p, st) : launch_attn_v2_one<D, NW, FMT, CAUSAL, false, false>(p, st);
#endif
}

// One translation unit per operand format (build.py: -DQATTN_ONLY_FMT=0|1); without the macro the file provides both.
template <int FMT>
static int launch_attn_v2_fmt(const AttnParams& p, int causal, int scale_mode, hipStream_t st) {
    return causal ? launch_attn_v2_t<128, 8, FMT, true>(p, scale_mode, st) : launch_attn_v2_t<128, 8, FMT, false>(p, scale_mode, st);
}
#if !defined(QATTN_ONLY_IN16) || QATTN_ONLY_IN16 == 2
#if !defined(QATTN_ONLY_FMT) || QATTN_ONLY_FMT == 0
int QATTN_V2_ENTRY(launch_attn_v2_e4m3)(const AttnParams& p, int causal, int scale_mode, hipStream_t st) { return launch_attn_v2_fmt<QATTN_FMT_E4M3>(p, causal, scale_mode, st); }
#endif
#if !defined(QATTN_ONLY_FMT) || QATTN_ONLY_FMT == 1
int QATTN_V2_ENTRY(launch_attn_v2_e5m2)(const AttnParams& p, int causal, int scale_mode, hipStream_t st) { return launch_attn_v2_fmt<QATTN_FMT_E5M2>(p, causal, scale_mode, st); }
#endif
#endif
#if !defined(QATTN_ONLY_IN16) || QATTN_ONLY_IN16 == 3
template <int FMT>
static int launch_attn_v2_f16_fmt(const AttnParams& p, int causal, hipStream_t st) {
    return causal ? launch_attn_v2_one<128, 8, FMT, true, false, true, true, QATTN_FMT_FP16>(p, st)
                  : launch_attn_v2_one<128, 8, FMT, false, false, true, true, QATTN_FMT_FP16>(p, st);
}
#if !defined(QATTN_ONLY_FMT) || QATTN_ONLY_FMT == 0
int QATTN_V2_ENTRY(launch_attn_v2_f16_e4m3)(const AttnParams& p, int causal, hipStream_t st) { return launch_attn_v2_f16_fmt<QATTN_FMT_E4M3>(p, causal, st); }
#endif
#if !defined(QATTN_ONLY_FMT) || QATTN_ONLY_FMT == 1
int QATTN_V2_ENTRY(launch_attn_v2_f16_e5m2)(const AttnParams& p, int causal, hipStream_t st) { return launch_attn_v2_f16_fmt<QATTN_FMT_E5M2>(p, causal, st); }
#endif
#endif

}  // namespace qattn
