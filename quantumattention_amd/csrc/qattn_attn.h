// qattn_attn.h -- parameters and helpers shared by the attention kernel variants (gfx950 only).
#pragma once
#include "qattn_common.h"

namespace qattn {

constexpr int kWaves = 8;                    // waves per workgroup (2 per SIMD)
constexpr int kThreads = kWaves * 64;
constexpr int kQPerWave = 32;                // query rows per wave (MFMA N)
constexpr int kQPerWG = kWaves * kQPerWave;  // 256
constexpr float kPShift = 5.0f;              // P' = P * 2^5 keeps small probabilities above the e4m3 subnormals
constexpr float kRescaleThr = 3.0f;          // log2 units: P' <= 2^(5+3) = 256 < 448 (e4m3 max)
constexpr float kPShiftByte = 5.0f;          // byte-exponential mode: P' = P * 2^5 and a deferred-rescale threshold of 3:
constexpr float kRescaleThrByte = 3.0f;      //   P' <= 2^8 -> byte <= 120 < 0x7e; a tighter threshold (1) made the fix-up frequent
constexpr float kByteBias = -0.3f;           // centres the (1+m/8 >= 2^(m/8)) mantissa error of the byte exponential
// The byte exponential's e4m3 value is, on average over the mantissa positions, exp(0.01353) of the true 2^x (with kByteBias above): numerator
// and denominator of the output see the same weights, so O is unaffected, but the log-sum-exp of the fused entry is formed from the sum of
// those weights and carries the offset -- measured +0.01353 +- 0.001 (std) over 10^5 rows, S = 2048 .. 8192, score spread 0.5 .. 1.2
// (profiles/r06/lse_signed_error.log); the epilogue subtracts it, which leaves |error| <= 1.5e-2 on every measured row (typically 4e-3).
constexpr float kByteLseBias = 0.01353f;
constexpr int kDynMinRounds = 24;           // non-causal launches with this many query blocks per workgroup draw them dynamically (see launch_attn_v2_chk)
constexpr int kTwoTermKeys = 1024;           // query blocks that see fewer keys than this use hi+lo (two-term) fp8 P from the start
// One-term rows are re-done with two-term P when R = l / p_max (the inverse of the row's largest softmax weight) ends below
// this: the error a single e4m3-rounded weight w contributes is about w * 2^-4 * |v - O|  
constexpr float kPeakR0 = 24.0f;
// ... and when its effective key count N_eff = l^2 / sum P'^2 ends below this.  R bounds the LARGEST weight; K similar weights
// that carry a row have R ~ K and a one-term error ~ eps_rms |v - O| / sqrt(K): 0.026 at K = 40, 0.017 at K = 100, 0.0125 at
// K = 160, 0.010 at K = 256 (tools/models/sim_heavy.py; 2^-6 = 0.0156).  N(0,1) rows sit at n / e: 1500 at n = 4096, 390 at 1024,
// where a row with R >= 24 is never below 192 (3e-4 of the rows at n = 1024).
constexpr float kPeakNeff = 192.0f;
// the e4m3 byte of P' read as e5m2 is 0.444 .. 0.5 of P'^2 over the eight mantissa values (mean 0.480); the MFMA that sums
// it stands for kNeffByteRatio * sum P'^2
constexpr float kNeffByteRatio = 0.472f;
constexpr float kNeffByteLow = 0.444f;     // ... and its lower end (mantissa 100b: 1.0 / 1.5^2)
constexpr float kCrushMean = 0.0625f;      // a one-term row's other keys must average at least this P' (four times e4m3's smallest normal), row_is_peaked
// Which blocks START in two-term mode: those whose rows are predicted to end below kPeakR0 anyway.  For scores ~ N(0, var)
// over n keys the row sum is about n exp(var / 2) and the largest term sits about z standard deviations out, so the smallest
// R in a block is about n exp(var / 2 - z sqrt(var)); z (AttnParams::peak_z = 1/2 + ln(kTwoTermKeys / kPeakR0) = 4.25) makes
// that meet kPeakR0 at n = kTwoTermKeys for unit variance -- round 1's key-count rule is this rule at var = 1, which is what
// a caller without the pre-pass's moments gets.  The fused step passes every head's sum of squares (ssq_q, ssq_k), and
// var = sm_scale^2 sum_d E[q_d^2] E[k_d^2] ~= sm_scale^2 ssq_q ssq_k / (Sq Skv D): a head with score spread 2 goes
// straight to two-term P instead of sweeping once in vain (AUTO on q x 2 data: 2.9x -> 1.9x the one-term time).  The
// prediction only picks the starting mode; the R test after a one-term sweep stays the arbiter of accuracy.  Estimates
// below kVarDeadband count as 1: unit-variance data takes the same decisions with and without the moments, so the fused step
// and the quantise-then-attend sequence of C calls stay bit-identical there.
constexpr float kVarDeadband = 1.5f;
// A head (or a wave's sample of scores) that IS wide is judged with a larger z: what decides a block's fate is whether more
// than max_rescue of its 32-row groups hold a flagged row, and with the row-to-row variation of |q| that happens from a
// score spread of about 1.2 on at n = 4096 (measured: q x 1.15 a fifth of the blocks repeat, q x 1.25 nearly all):
// 4096 exp(1.2^2 / 2 - z 1.2) = 24 at z = 4.9.
constexpr float kPeakZWide = 4.9f;
constexpr int kVxWords = 256;                // V chunk scale bytes a block keeps in LDS (block-scaled V: Skv <= 16384)
__device__ inline float predicted_r(float nkeys, float var, float z) {
    // (opaque 0.5: left to the compiler, {0.5, z} becomes a loop-invariant register pair of a v_pk_mul_f32 that is hoisted to the
    // top of the persistent kernels' block loop and spilled -- the kernels sit at the 256-register limit)
    float half = 0.5f;
    asm volatile("" : "+v"(half));
    return nkeys * __expf(half * var - z * sqrtf(var));
}
// "So many rows of a block will end peaked that the block should run two-term from the start": rows, gathered across the waves of a
// block, are the unit of the rescue (rescue_pass), and recomputing g groups of 32 costs about 0.28 g sweeps against 1.8 for the two-term
// sweep -- worth it up to about a third of the block's rows.  For scores ~ N(0, var) over n keys a row ends with R < kPeakR0 when its
// largest score sits more than z* standard deviations out, n exp(var / 2 - z* sigma) = kPeakR0 (predicted_r), which happens to a
// fraction n (1 - Phi(z*)) of the rows; that fraction is 0.35 at z* = sqrt(2 ln n) - 0.38 (3.34 / 3.70 / 4.03 at n = 1024 / 4096 /
// 16384).  Second criterion: the effective key count n exp(-var) of such rows falls below kPeakNeff (with a margin) -- then EVERY row
// is flagged whatever its largest weight (score spread >= 1.7 at n = 4096).
constexpr float kNeffStartMargin = 1.25f;
__device__ inline bool many_rows_peaked(float nkeys, float var) {
    const float z_many = sqrtf(2.0f * __logf(fmaxf(nkeys, 2.0f))) - 0.38f;
    return predicted_r(nkeys, var, z_many) < kPeakR0 || nkeys * __expf(-var) < kNeffStartMargin * kPeakNeff;
}
// a head's sum of squares from its partial sums, the same value in every lane of every wave (fixed order: lane l adds
// l, l + 64, ...; then a fixed reduction tree)
__device__ inline float sum_partials(const float* part, int n, int lane) {
    float t = 0.0f;
    for (int i = lane; i < n; i += 64) t += part[i];
    return wave_allsum(t);
}

// Both heads' sums at once, branch-free (n <= 256 = 4 words per lane, clamped indices, zeros beyond n: the same additions in the
// same order as sum_partials): eight loads in flight and one wait instead of a wait per loop trip and per head.
__device__ inline void sum_partials_pair(const float* pa, const float* pb, int n, int lane, float& sa, float& sb) {
    float a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { const int i = min(lane + 64 * k, n - 1); a[k] = pa[i]; b[k] = pb[i]; }
    float ta = 0.0f, tb = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; k++) { const bool in = lane + 64 * k < n; ta += in ? a[k] : 0.0f; tb += in ? b[k] : 0.0f; }
    sa = wave_allsum(ta);
    sb = wave_allsum(tb);
}

struct AttnParams {
    const unsigned char* q;
    const unsigned char* k;
    const unsigned char* v;
    void* out;
    float* lse;
    const float* sq;
    const float* sk;
    const float* sv;
    int B, Hq, Hkv, Sq, Skv;
    int nqb;         // query blocks (of kQPerWG rows) per head
    int nchunks;     // 64-key chunks per head
    int out_fmt;
    int xcd_remap;   // 1: each XCD gets a contiguous range of heads
    int causal_group;  // causal: heads per longest-first group within an XCD (1 = head after head)
    int sched_zeroed;         // the hand-out counters were cleared by the kernel before this launch (fused step)
    int dyn_min_rounds;       // non-causal launches with at least this many blocks per workgroup use the dynamic hand-out too
    int risky_lo, risky_hi;   // causal, AUTO: query blocks [lo, hi) of a head go FIRST (map_block); lo = hi: plain longest-first
    int tail_lo;              // causal, XCD-aware hand-out: the query blocks qb < tail_lo of ALL of an XCD's heads are handed out after everything else (map_block); 0: each group's own
    float sm_log2e;  // sm_scale * log2(e)
    int precision;   // QATTN_PRECISION_*
    int two_term_keys;  // kTwoTermKeys (a development switch can change it)
    int n_two;       // leading query blocks per head that start in two-term mode (set by the launcher)
    float peak_r0;
    float peak_neff;             // kPeakNeff (0 with peak_r0 = 0)
    const float* ssq_q;          // fused step, head-wise: partial sums of squares of every q head [B*Hq][ssq_stride] and k
    const float* ssq_k;          //   head [B*Hkv][ssq_stride], ssq_n of them valid per head (else nullptr)
    int ssq_n, ssq_stride;       // (a caller-supplied sum of squares per head: one entry, stride 1)
    float peak_z;                // see predicted_r
    float var_mul;               // score variance of head (bh, kvh) ~= sum(ssq_q[bh]) * sum(ssq_k[kvh]) * var_mul
    int total_blocks;            // B * Hq * nqb (set by the launcher); the grid may be smaller: workgroups walk blocks bid, bid + gridDim.x, ...
    int persistent;              // one workgroup per CU instead of one per block
    // Dynamic block hand-out of the hand-scheduled kernel's causal launches (qattn_attn_v2.hip; nullptr: static).  `sched` lives in
    // the call's workspace and is zeroed by the launcher every call: the workgroups of a persistent launch draw their next
    // block from per-XCD counters (SchedState below).
    struct SchedState* sched;
    int sched_nq;                // block counters in use: 8 (one per XCD label blockIdx.x & 7) with xcd_remap, else 1
    const unsigned* vexp;        // fused step with a block-scaled V (else nullptr): E8M0 byte of every 64-key V chunk, [B*Hkv][ssq_stride]
    int max_rescue_rows;         // D = 128 kernel: more peaked ROWS than this in a 256-row block: the block is redone in two-term mode (fewer: gathered and rescued)
    int max_rescue;              // more peaked 32-row groups than this in a 256-row block: the block is redone in two-term mode   // > 0: one-term blocks with a row of R < peak_r0 are repeated in two-term mode (QATTN_PRECISION_AUTO)
    unsigned* flags; // templated kernel (qattn_attn_v4.hip): one word per (head, 32-row group), set by the one-term launch
    long lse_stride; // floats between the LSE rows of consecutive (b, h)
    float lse_mul;   // 1 (natural log-sum-exp) or -sqrt(D) (QATTN_LSE_REFERENCE)
    // The ORIGINAL 16-bit V, row-major [B,Hkv,Skv,D] (else nullptr): what pv16_block_pass (qattn_pv16.h) attends -- every block of a
    // call with v_fmt = QATTN_FMT_BF16 / _FP16, and in the fused step the blocks that see fewer than two_term_keys keys (bf16)
    const unsigned char* v16;
    const unsigned char* q16;      // fused step: the 16-bit (bf16) Q tensor, quantised row by row in the kernel prologue (else nullptr)
    const unsigned* q_amax_part;   // fused step: abs-max words (fp32 bits) of every q head, [B*Hq][amax_stride], amax_n valid per head:
    int amax_n, amax_stride;       //   the abs-max pass's per-block words, or ONE caller-supplied word per head (producer hand-off)
    int vexp_stride;               // words between the V chunk scale bytes of consecutive kv heads
    float* sq_out;                 // fused step: scale_q [B,Hq] is written by the attention kernel
    int q_numerics;
    unsigned long long* stamp_buf;   // measurement entry only (else nullptr): {shader cycles, 100 MHz ticks} of every wave's KV sweep
    // Debug output of the fused entry (else nullptr): one byte per query row [B,Hq,Sq], pre-filled with QATTN_PATH_ONE_TERM by the call; every
    // pass that is NOT the one-term fp8-V sweep overwrites the rows it stores: QATTN_PATH_TWO_TERM (two-term fp8 P on the fp8 V: rescued
    // rows, two-term blocks of the templated kernel) or QATTN_PATH_V16 (16-bit P on the caller's 16-bit V: early blocks, blocks on the
    // 16-bit-V pass, severely peaked rescued rows).  The one-term sweeps never touch it.
    unsigned char* path;
    // The caller's 16-bit Q and V as the attention kernels read them (q16 / v16 above) may be STRIDED views of [B,H,S,D] -- D innermost and
    // dense, e.g. the transposed view of a [B,S,H,D] projection output (the reference's launcher copies such a V: tk/attention.py:419-421).
    // Byte strides of batch, head and row; dense: {H S D 2, S D 2, D 2}.  (attention_impl, qattn_api.hip)
    long q16_bs, q16_hs, q16_rs;
    long v16_bs, v16_hs, v16_rs;
    // `out` may be a strided [B,Hq,Sq,D] view too (D innermost, dense) -- e.g. the transposed view of a [B,Sq,Hq,D] buffer, which a caller
    // then reshapes to [B,Sq,Hq D] for its output projection without a copy.  Byte strides of batch, head, row; dense: {Hq Sq D 2, Sq D 2, D 2}.
    long o_bs, o_hs, o_rs;
};

// QATTN_STRIDED16 (a per-translation-unit build switch, build.py): 1 (default) = the kernels of this unit address the 16-bit Q / V and the
// output through the strides above; 0 = dense arithmetic with compile-time row sizes.  The hand-scheduled kernel is built BOTH ways
// (qattn_attn_v2_* / qattn_attn_v2_*_sv units; launch_attn_v2 picks by the call's strides): its dense instantiations keep, instruction for
// instruction, the code they had before strides existed -- any change to that kernel's source moves its schedule by +-0.5 % (three
// formulations of the strided addressing measured +0.3 / +0.7 / +1.3 % on the dense C2 step: profiles/r06/ab_strided_*.log).
#ifndef QATTN_STRIDED16
#define QATTN_STRIDED16 1
#endif
constexpr bool kStrided16 = QATTN_STRIDED16 != 0;

// first byte of row `row` of head (b, h) of the 16-bit Q (row_bytes: its dense size) / of kv head (b, hkv) of the 16-bit V
__device__ __forceinline__ const unsigned char* q16_row(const AttnParams& p, int b, int h, long bh, int row, int row_bytes) {
    if (kStrided16) return p.q16 + (long)b * p.q16_bs + (long)h * p.q16_hs + (long)row * p.q16_rs;
    return p.q16 + (bh * p.Sq + row) * row_bytes;
}
__device__ __forceinline__ const unsigned char* v16_head(const AttnParams& p, int b, int hkv, long kv_head, int row_bytes) {
    if (kStrided16) return p.v16 + (long)b * p.v16_bs + (long)hkv * p.v16_hs;
    return p.v16 + kv_head * (long)p.Skv * row_bytes;
}
// bytes between consecutive rows of the 16-bit V
__device__ __forceinline__ long v16_row_stride(const AttnParams& p, int row_bytes) { return kStrided16 ? p.v16_rs : (long)row_bytes; }
// byte offset of output row `row` of head bh = b Hq + h (store_o_rows); row_bytes: the dense size of a row
__device__ __forceinline__ long out_head_offset(const AttnParams& p, int b, int h) { return kStrided16 ? (long)b * p.o_bs + (long)h * p.o_hs : 0L; }
__device__ __forceinline__ long out_row_offset(const AttnParams& p, long bh, int row, int row_bytes) {
    if (kStrided16) {
        const int bhi = (int)bh, b = bhi / p.Hq;   // (32-bit: B Hq < 2^31)
        return out_head_offset(p, b, bhi - b * p.Hq) + (long)row * p.o_rs;
    }
    return (bh * p.Sq + row) * row_bytes;
}
// the same with the head's offset at hand (the hand-scheduled kernel's sweeps: no division in the epilogue)
__device__ __forceinline__ long out_row_offset(const AttnParams& p, long o_head, long bh, int row, int row_bytes) {
    return kStrided16 ? o_head + (long)row * p.o_rs : (bh * p.Sq + row) * row_bytes;
}

// ---------------------------------------------------------------------------------------------------------
// Block hand-out inside one persistent causal launch of the D = 128 kernel.  The only datum that travels between workgroups is
// a block number drawn from a counter (relaxed agent-scope atomics; nothing a block reads is written in the launch, so no
// release / acquire is involved):
//   next[x]   blocks handed out from queue x beyond the first gridDim.x / nq (every workgroup starts on block blockIdx.x);
//             x = blockIdx.x & 7 labels the workgroups that share an XCD, so a head's blocks keep their K / V in one L2
// One 32-byte header to zero per launch.
// Measured and dropped (round 3, profiles/r03/sched_queue.md): a queue of the 32-row groups to rescue, emptied by workgroups
// that have run out of blocks.  A rescue away from the XCD whose L2 holds the head's K / V is 2-3x slower (latency-bound,
// chunk by chunk), a rescue cannot be split, and the workgroup that found the group is as fast as any idle one of its XCD:
// deferring every rescue to the end of a causal launch cost +40 %, handing over only the groups of a workgroup's last block
// +12 % (S = 4096), against rescues on the spot.
// ---------------------------------------------------------------------------------------------------------
struct SchedState {
    unsigned next[8];
};
inline size_t sched_bytes() { return sizeof(SchedState); }   // a multiple of 16

// Zeroes the scratch words a launch starts from (block hand-out counters, peaked-group flags).  A kernel, not hipMemsetAsync: as
// a memset NODE of a captured graph the 32-byte fill faulted on the second replay (ROCm 7.2, found by the graph test on a
// causal launch with more blocks than CUs); kernel nodes replay fine.
__global__ void zero_words_kernel(unsigned* w, long n);
inline hipError_t zero_words(unsigned* w, long n, hipStream_t st) {
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, n);
    return hipGetLastError();
}

#define QATTN_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
// the next block of queue x, else of another queue (a workgroup that has finished its XCD's blocks helps elsewhere); -1: none left.
// Two halves, so that a caller can put work between the request and its use: sched_draw_issue returns the raw ticket of queue x.
__device__ inline unsigned sched_draw_issue(SchedState* s, int x) { return __hip_atomic_fetch_add(&s->next[x], 1u, QATTN_RLX_AGENT); }
__device__ inline int sched_draw_finish(SchedState* s, int nq, int x, int bpq, int first, unsigned ticket) {
    // (unsigned comparisons: whatever the counters hold, a block number outside [0, bpq * nq) is never handed out)
    const unsigned idx = ticket + (unsigned)first;
    if (idx < (unsigned)bpq) return (int)idx * nq + x;
    unsigned nx[8];
#pragma unroll
    for (int k = 0; k < 8; k++) nx[k] = __hip_atomic_load(&s->next[k & (nq - 1)], QATTN_RLX_AGENT);   // all in flight together
#pragma unroll
    for (int k = 1; k < 8; k++) {
        const int xx = (x + k) & (nq - 1);
        if (k >= nq || nx[xx] + (unsigned)first >= (unsigned)bpq) continue;
        const unsigned j = __hip_atomic_fetch_add(&s->next[xx], 1u, QATTN_RLX_AGENT) + (unsigned)first;
        if (j < (unsigned)bpq) return (int)j * nq + xx;
    }
    return -1;
}
__device__ inline int sched_next_block(SchedState* s, int nq, int x, int bpq, int first) {
    return sched_draw_finish(s, nq, x, bpq, first, sched_draw_issue(s, x));
}

// The verdict on one row at the end of a one-term sweep.  l = sum P', l2 = sum P'^2 (BYTE: kNeffByteRatio of
// it), r_inv_pmax = 1 / P'_max, all in the row's final reference.
//   * N_eff = l^2 / l2 < peak_neff: many similar weights, a statistical error that R does not see.
//   * R = l / P'_max < peak_r0: one weight so large that its own rounding may exceed the budget -- UNLESS that weight is exact.
//     A key whose score beat the row's reference max by more than the deferred-rescale threshold took the fix-up branch and
//     became the reference: it was exponentiated at x = shift exactly, its byte (or RNE e4m3 value) is exactly 2^shift, and
//     no later rescale touched the row (m_run == m_true from then on), so its contribution carries no rounding error at
//     all.  The row is then judged by its REST: with the top key taken out of both sums, l_r^2 / l2_r >= peak_r0^2 says that
//     no other weight exceeds 1 / peak_r0 -- the same guarantee the R test gives an ordinary row.  On N(0,1) data nearly
//     every row with R < 24 is one outlier key of exactly this kind (tools/models/sim_exact_top.py: 100 % at n = 4096, 77 % at 2048).
template <bool BYTE, bool NEFF>
__device__ __forceinline__ bool row_is_peaked(const AttnParams& p, float l, float l2, float r_inv_pmax, bool top_is_reference, float nkeys) {
    static_assert(kPShift == 5.0f && kPShiftByte == 5.0f, "p_top below is 2^shift");
    bool peaked = l * r_inv_pmax < p.peak_r0;
    if (NEFF) {
        constexpr float ratio = BYTE ? 1.0f / kNeffByteRatio : 1.0f;
        constexpr float p_top = 32.0f;                                          // 2^shift, and its square as the l2 sum holds it:
        constexpr float p_top2 = BYTE ? 0.5f * p_top * p_top : p_top * p_top;   //   the byte of 2^integer read as e5m2 is exactly half the square
        const float l_r = l - p_top, l2_r = fmaxf(l2 - p_top2, 0.0f);
        const bool rest_flat = top_is_reference && l_r * l_r >= p.peak_r0 * p.peak_r0 * ratio * l2_r;
        peaked = (peaked && !rest_flat) || l * l < p.peak_neff * ratio * l2;
        // The same two guarantees for a row whose top key is exact, from the rest's moments AND the number of keys the row saw:
        // with w2 the largest remaining weight, sum_rest w^2 >= w2^2 + (A - w2)^2 / (n - 2) (A = 1 - w_top: Cauchy-Schwarz on
        // the n - 2 others), and the right side grows with w2 beyond A / (n - 1).  So sum_rest w^2 < t^2 + (A - t)^2 / (n - 2)
        // with t = 1 / peak_r0 proves w2 < t; and for n > 1.5 peak_neff + 2 that bound is itself below 1 / peak_neff, the
        // statistical budget (the exact top key adds no error to either).  The rule above (t^2 alone on the right) refuses every
        // flat row that sees fewer than e peak_r0^2 = 1566 keys -- the band where all rescues of a flat causal launch fell.
        // The byte estimate of sum P'^2 enters with its worst case (1 / kNeffByteLow of the e5m2 sum), not its mean.
        static_assert(kPeakR0 * kPeakR0 == 3.0f * kPeakNeff, "the key-count floor below is 1 / (1 / neff - 1 / r0^2) + 2");
        if (top_is_reference && peaked && nkeys > 1.5f * p.peak_neff + 2.0f) {
            constexpr float worst = BYTE ? 1.0f / kNeffByteLow : 1.0f;
            const float t = l / p.peak_r0, rest = l_r - t;
            peaked = !(rest > 0.0f && worst * l2_r < t * t + rest * rest / (nkeys - 2.0f));
        }
        // Dynamic range.  P' lives between the reference 2^5 and e4m3's smallest normal 2^-6; a row whose OTHER keys sit, on average,
        // within two binades of that floor loses them to the subnormals (the byte formula is not even monotone-exact there) and
        // to zero -- however flat they are among themselves, and the sums above, made of the crushed bytes, cannot show it.  The R
        // test used to imply a floor (l >= 24 x 32 spread over n keys); a row excused from it by its exact top key has none,
        // found by tools/fuzz_parity.py on rows with one key 8 .. 10 nats above the rest (errors of 0.1 .. 0.3).  Model
        // (tools/models/sim_crushed_rest.py): mean rest P' >= 0.05 keeps the one-term error below 0.012, < 0.04 does not.
        peaked = peaked || l_r < kCrushMean * nkeys;
    }
    return peaked;
}

// PV product with a block-scaled V: `scale_word` carries the E8M0 byte (e + 127) of the V chunk in A in byte 0 (one value for
// both 32-wide K blocks of every row here; profiles/r02_mfma_scale_probe.log) and 127 = 2^0 for B = P in byte 1 -- the
// instruction takes its two scales from vector registers only, op_sel picks the byte, so one register serves both (a
// separate register holding the constant 127 cost the tightest instantiations a spill).  VS = false: the plain product.
constexpr int kScaleWordOne = (127 << 8) | 127;   // A x 2^0, B x 2^0
__device__ __forceinline__ int vscale_word(unsigned e8m0_byte) { return (int)((e8m0_byte & 0xffu) | (127u << 8)); }
template <int CBSZ, int BLGP, bool VS>
__device__ inline v16f mfma_pv(v8i a, v8i b, v16f c, int scale_word) {
    if constexpr (VS) return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, CBSZ, BLGP, 0, scale_word, 1, scale_word);
    else return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, CBSZ, BLGP, 0, 0, 0, 0);
}

template <int CBSZ, int BLGP>
__device__ inline v16f mfma_f8(v8i a, v8i b, v16f c) {
    // scale operands 0 -> the unscaled v_mfma_f32_32x32x64_f8f6f4 (implicit scale 1.0; profiles/r01_mfma_probe.log)
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, CBSZ, BLGP, 0, 0, 0, 0);
}

// two floats -> two bf16 / fp16 in one dword (round to nearest even; v_cvt_pk_bf16_f32 / v_cvt_pkrtz is NOT used for fp16)
__device__ inline unsigned pack2_bf16(float a, float b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const b2 h = __builtin_convertvector(f2{a, b}, b2);
    unsigned u;
    __builtin_memcpy(&u, &h, 4);
    return u;
}
__device__ inline unsigned pack2_f16(float a, float b) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 h = {(_Float16)a, (_Float16)b};
    unsigned u;
    __builtin_memcpy(&u, &h, 4);
    return u;
}

// One float at a workgroup-uniform address through the scalar cache.  The compiler only emits scalar loads for memory it can
// prove unwritten by the kernel; a VECTOR load of a uniform value that some path never consumes stays "pending" on its
// register, and the next write to that register then waits for every outstanding vector-memory operation -- stores included.
__device__ __forceinline__ float scalar_load_f32(const float* p) {
    float v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() is fence + barrier, and the fence drains the wave's
// global stores too (s_waitcnt vmcnt(0)): at the end of a query block that is the write acknowledgement of the block's O rows
// (1 - 2 us on this chip) in front of a barrier whose only job is to hand over the K/V ring, the Q slots and the vote words.
// gfx950 backs a barrier off while memory operations are outstanding, so only the LDS counter has to be waited for.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// The few words the waves of a workgroup pass each other at the end of a block (votes, the next block's number), written and
// read through asm: in front of every LDS access of its own the compiler waits for ALL outstanding vector-memory operations
// (it cannot tell the access from the LDS-DMA destinations, and stores share the counter) -- right behind the O stores that
// is their write acknowledgement again.  The K/V ring's DMA of the block has been waited for long before (kv_sweep).
__device__ __forceinline__ unsigned lds_addr(const volatile void* p) {
    return (unsigned)(unsigned long)(const volatile __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ void lds_write_word_raw(volatile unsigned* p, unsigned v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(lds_addr(p)), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_write_byte_raw(volatile void* p, unsigned v) {
    asm volatile("ds_write_b8 %0, %1" ::"v"(lds_addr(p)), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned lds_read_word_raw(const volatile unsigned* p) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(lds_addr(p)) : "memory");
    return v;
}
__device__ __forceinline__ void lds_read_8words_raw(const volatile unsigned* p, v4i& a, v4i& b) {   // p: 16-byte aligned
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(lds_addr(p)) : "memory");
}

// v_max3_f32 through asm: on MFMA results the compiler otherwise adds a canonicalising v_max_f32 x, x, x per chain.  Interleave
// THREE independent chains: hipcc pads wait states between an asm statement and a VALU that reads its output unless two other
// instructions sit in between.
// HAZARD: the compiler does NOT count the MFMA -> VALU read wait states for an operand of inline asm (it does for its own
// instructions).  A caller that feeds MFMA results into max3_raw must know they have landed: the hand-scheduled kernel reads a
// score tile a whole pipeline step (>= 2 later MFMAs) after the MFMAs that wrote it; everything else goes through
// max32_after_mfma below.
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// The maximum of the 32 scores a lane holds of one 64-key chunk (s0, then s1 = the accumulators of two QK^T MFMAs issued in this
// order).  The first touch is a plain fmaxf on the YOUNGER tile: the compiler waits out the MFMA latency in front of it, and every
// asm chain starts from its result, so none of them can be scheduled ahead of it.
__device__ __forceinline__ float max32_after_mfma(const v16f& s0, const v16f& s1) {
    const float first = fmaxf(s1[14], s1[15]);
    float a = max3_raw(first, s0[0], s0[1]), b = max3_raw(first, s0[2], s0[3]), c = max3_raw(first, s0[4], s0[5]);
    a = max3_raw(a, s0[6], s0[7]); b = max3_raw(b, s0[8], s0[9]); c = max3_raw(c, s0[10], s0[11]);
    a = max3_raw(a, s0[12], s0[13]); b = max3_raw(b, s0[14], s0[15]); c = max3_raw(c, s1[0], s1[1]);
    a = max3_raw(a, s1[2], s1[3]); b = max3_raw(b, s1[4], s1[5]); c = max3_raw(c, s1[6], s1[7]);
    a = max3_raw(a, s1[8], s1[9]); b = max3_raw(b, s1[10], s1[11]); c = max3_raw(c, s1[12], s1[13]);
    return max3_raw(a, b, c);
}

__device__ inline v8i lds_read_frag(const unsigned char* base) {
    // two ds_read_b128: pieces [half=0] and [half=1] are 512 bytes apart
    v4i lo = *reinterpret_cast<const v4i*>(base);
    v4i hi = *reinterpret_cast<const v4i*>(base + 512);
    v8i r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// Epilogue shared by the attention kernels: O^T accumulators (query on the lane, d in the registers) -> normalised 16-bit
// rows.  The lanes of the two half-waves hold the 8-byte halves of each 16-byte piece of a row; one v_permlane32_swap per
// dword pairs them up so that every lane stores 16 contiguous bytes (half the store instructions of the 8-byte form).
template <int MB>
__device__ __forceinline__ void store_o_rows(void* out, int out_fmt, const v16f (&o)[MB], float inv, long row_offset, int hh, bool valid) {
    unsigned char* op = reinterpret_cast<unsigned char*>(out) + row_offset + (hh << 4);   // row_offset: out_row_offset() bytes
    const bool bf = out_fmt == QATTN_FMT_BF16;
#pragma unroll
    for (int m = 0; m < MB; m++)
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            unsigned a[2], b[2];  // this lane's 4 + 4 elements of column groups j and j + 1
#pragma unroll
            for (int i = 0; i < 2; i++) {
                a[i] = bf ? pack2_bf16(o[m][4 * j + 2 * i] * inv, o[m][4 * j + 2 * i + 1] * inv)
                          : pack2_f16(o[m][4 * j + 2 * i] * inv, o[m][4 * j + 2 * i + 1] * inv);
                b[i] = bf ? pack2_bf16(o[m][4 * j + 4 + 2 * i] * inv, o[m][4 * j + 5 + 2 * i] * inv)
                          : pack2_f16(o[m][4 * j + 4 + 2 * i] * inv, o[m][4 * j + 5 + 2 * i] * inv);
                const auto sw = __builtin_amdgcn_permlane32_swap(a[i], b[i], false, false);
                a[i] = sw[0]; b[i] = sw[1];
            }
            // lanes 0..31: [own group j | upper half's group j] = columns 32m + 8j .. +7; lanes 32..63: the next 8 columns
            // (the exchange runs with every lane active; only the store is predicated on the row being inside the tensor)
            // (non-temporal stores here cost the step 3-5 %, non-temporal loads of the 16-bit Q rows 2-3 % -- profiles/r03/ab_nt_loads.log)
            if (valid) *reinterpret_cast<v4i*>(op + (32 * m + 8 * j) * 2) = v4i{(int)a[0], (int)a[1], (int)b[0], (int)b[1]};
        }
}

template <int N>
__device__ inline void wait_vmcnt() {
    static_assert(N == 0 || N == 1 || N == 2 || N == 4, "unsupported vmcnt");
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    if (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
}

// CUs of the current device (cached per device ordinal) and its XCD count.  The block -> (head, query block) maps below and
// the persistent grids assume the SPX-mode chip: 8 XCDs of 32 CUs, workgroups dealt round-robin over them.  A partition of
// the chip (CPX / DPX / QPX: 32 / 128 / 64 CUs) or another part reports a different CU count; the launchers then use the
// plain maps (xcd_remap = 0), which are correct anywhere -- the XCD arithmetic is a speed assumption, never a correctness one.
inline int cu_count() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        cached[dev] = n > 0 ? n : -1;
    }
    return cached[dev] > 0 ? cached[dev] : 0;
}
constexpr int kCusPerXcd = 32;
inline int xcd_count() { const int c = cu_count(); return c > 0 && c % kCusPerXcd == 0 ? c / kCusPerXcd : 1; }

// block -> (head, query block).  Blocks b and b+8 share an XCD (round-robin dispatch; a speed assumption only):
// each XCD gets a contiguous range of heads so the heads it works on keep their K/V in its private 4 MiB L2.
// Causal: query blocks differ 16:1 in work, so within an XCD the heads are taken in groups of kCausalHeadGroup and a
// group's blocks are issued heaviest first ACROSS its heads (longest-processing-time order: the launch ends on the lightest
// blocks instead of on a late heavy one), as long as the group's K + V fit kCausalGroupBytes of the XCD's L2.
constexpr int kCausalHeadGroup = 4;
#ifndef QATTN_CAUSAL_GROUP_BYTES
#define QATTN_CAUSAL_GROUP_BYTES (2u << 20)   // (a build knob for tools/ab.py variants)
#endif
constexpr size_t kCausalGroupBytes = QATTN_CAUSAL_GROUP_BYTES;
// Causal order of a head's query blocks: longest first (LPT), except that the blocks most likely to need a rescue go before
// everything else.  With AUTO the blocks whose first row sees fewer than kTwoTermKeys keys start in two-term mode (no rescue
// possible), the ones just above that line (risky_lo <= qb < risky_hi: rows that see 1 .. 2 kTwoTermKeys keys) are where the
// one-term sweep meets borderline rows -- on N(0,1) data every rescue of a causal launch falls there.  Under plain LPT those
// short blocks come last, so their rescues (about 20 us each, a third of a long block) landed on the launch's tail, where
// no other work is left to balance them: workgroups idled 33 us on average before the end of a C3 launch, 16 us with FAST
// (dev work log).  Uncertain jobs first is the textbook order; the tail is then made of the cheap two-term
// blocks.
__device__ inline int causal_order(const AttnParams& p, int slot, int nqb) {
    const int hi = min(p.risky_hi, nqb), lo = min(p.risky_lo, hi);   // (a permutation of 0 .. nqb - 1 for any nqb)
    if (slot < hi - lo) return hi - 1 - slot;
    const int s2 = slot - (hi - lo);
    return s2 < nqb - hi ? nqb - 1 - s2 : lo - 1 - (s2 - (nqb - hi));
}
__device__ inline void map_block(const AttnParams& p, int bid, int nqb, bool causal, int& head, int& qb) {
    if (p.xcd_remap) {
        const int xcd = bid & 7, idx = bid >> 3, hpx = (p.B * p.Hq) >> 3;
        const int G = p.causal_group;
        if (causal && p.tail_lo > 0) {
            // Round 5: the lightest blocks of every head (qb < tail_lo: the rows that see fewer than kTwoTermKeys keys) are kept for the END of
            // the XCD's list.  Handed out group by group, the last group's 75 us blocks were taken when only two dozen blocks were left and the
            // launch ended on them with the other workgroups idle (35 us on average before the end of a C3 launch, 9 % of it: dev work log,
            // profiles/r05/work_log_c3.log); a list that ends on 4 tail_lo hpx blocks of 6 .. 25 us fills that time.  They no longer find their
            // head's K / V in L2 (they read at most a quarter of it).
            const int nh = nqb - p.tail_lo, heavy = hpx * nh;
            if (idx >= heavy) {   // heaviest first across the XCD's heads
                const int i2 = idx - heavy;
                qb = p.tail_lo - 1 - i2 / hpx;
                head = xcd * hpx + i2 % hpx;
                return;
            }
            const int per = G * nh, grp = idx / per, r = idx % per;
            head = xcd * hpx + grp * G + r % G;
            qb = causal_order(p, r / G, nqb);   // (slots < nqb - tail_lo: the risky blocks, then the long ones down to qb = tail_lo)
            return;
        }
        if (causal && G > 1) {
            const int per = G * nqb, grp = idx / per, r = idx % per;
            head = xcd * hpx + grp * G + r % G;
            qb = causal_order(p, r / G, nqb);
            return;
        }
        head = xcd * hpx + idx / nqb;
        qb = idx % nqb;
    } else {
        head = bid / nqb;
        qb = bid % nqb;
    }
    if (causal) qb = causal_order(p, qb, nqb);
}

// In-place fix-ups of a finished S^T chunk before its softmax (both rare or cheap, kept out of the hot block):
// token-wise key scales (inductor/kernels/attention.py:395) and the ragged-tail / causal-diagonal mask.
// RAGGED = false: the caller knows that the chunk lies inside the key range (every chunk but a head's last one); the test
// then needs nothing of AttnParams -- inside the hand-scheduled loop p.Skv was re-read from the kernel-argument segment every
// iteration (scalar registers are short there) and its s_waitcnt lgkmcnt(0) drained the operand reads in flight.
template <bool CAUSAL, bool TOKEN, bool RAGGED = true>
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
    const bool need_mask = (RAGGED && k0 + 64 > p.Skv) || (CAUSAL && k0 + 63 > q0);  // wave-uniform
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

// ---------------------------------------------------------------------------------------------------------
// Row-group rescue (QATTN_PRECISION_AUTO): recompute 32 query rows (one wave's group) with exact
// exponentials and two-term P, the key range split over the NW = 8 waves of a workgroup (each takes a contiguous run of
// 64-key chunks, reads the K / V fragments straight from global memory / L2 and keeps a partial {O, m, l}); the partials
// are merged pairwise through LDS (three rounds, 4 slots of SLOT bytes at `smem`) and wave 0 stores the rows.  Called
// from inside the D = 128 kernel (by whichever workgroup takes the queue item; the idle K/V ring holds the merge) and from
// rescue_groups_kernel for the templated kernel's launches; the Q^T fragments come from global memory through `qfrag`.
// About 0.3 of a 256-row block's sweep.
// ---------------------------------------------------------------------------------------------------------
constexpr int kMaxRescueWaves = 2;   // (templated kernel) more peaked 32-row groups than this in a 256-row block: the block is redone in two-term mode
// (fused step) flagged rows go to the 16-bit-V rescue (qattn_pv16.h rescue_rows16_at) when one of the block's flagged rows is SEVERELY
// peaked, R = 1 / w_max < kPeakR16: the fp8 V's rounding then reaches the output with a weight above 1/8 (about 2^-7 for unit-variance
// V).  Milder rows -- R between 8 and 24, all that flat data ever flags -- keep the two-term fp8 rescue, which takes two thirds of the time
// (a rescue at the end of a launch is its tail: with every rescue on the 16-bit V the C2 step was 1.6 % slower, C3 2.3 %,
// profiles/r05/ab_c{2,3}_all_rescues_16bit_vs_r4.log).
constexpr float kPeakR16 = 8.0f;
constexpr int kRescueSevere = 1 << 16;   // flag on attend_block's / run_block's row count
constexpr int kMaxRescueRows = 96;   // (D = 128 kernel) more peaked rows than this in a 256-row block: redone; else gathered into <= 3 groups of 32 and recomputed

__device__ __forceinline__ v8i gload_frag(const unsigned char* base) {
    const v4i lo = *reinterpret_cast<const v4i*>(base);
    const v4i hi = *reinterpret_cast<const v4i*>(base + 512);
    v8i r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

template <int D>
constexpr int rescue_slot_bytes() { return ((D / 32) * 16 + 2) * 64 * 4; }   // one wave's partial {O^T, m, l}

// The 32 rows are ANY rows of one head, one per lane pair (lanes l and l + 32): `row` (per lane; c and qfrag belong to it), of which
// the lanes with `store` set are written; row_lo / row_hi (wave-uniform) bound the rows for the causal chunk count and the mask test.
// The D = 128 kernel passes the flagged rows of a 256-row block, gathered from all its waves (rescue_pass); the wrapper below is the
// contiguous group r0 .. r0 + 31 of the templated kernel's rescue launch.  (The rows' QATTN_PATH_TWO_TERM codes are written by the CALLERS
// once this function's registers are dead: inside it the store cost the D = 256 rescue kernel a spilled register.)
template <int D, int NW, int QK_FMT, int V_FMT, bool CAUSAL, bool TOKEN, bool QLDS = false, bool VSCALE = false, typename QFrag>
__device__ __forceinline__ void rescue_rows_at(const AttnParams& p, unsigned char* smem, const unsigned char* kg, const unsigned char* vg,
                                               int row, bool store, int row_lo, int row_hi, int wave, int lane, long bh, long kv_head, float c,
                                               const float* skt, QFrag&& qfrag, const unsigned* vx = nullptr) {   // vx: the V chunks' scale bytes (LDS), nullptr: unscaled V
    static_assert(NW == 8, "three merge rounds");
    constexpr int CH = 64 * D, KS = D / 64, MB = D / 32;
    constexpr int SLOT = rescue_slot_bytes<D>();
    constexpr bool QPRE = D <= 128 && !QLDS;   // Q^T fragments held in registers (D = 256: re-fetched per chunk, registers go to
                                               // O^T; QLDS: the caller's qfrag reads LDS, cheap enough to repeat per chunk)
    constexpr int VB = MB > 4 ? 2 : MB;   // V fragments requested ahead of the PV MFMAs
    const int ql = lane & 31, hh = lane >> 5;
    const int r0 = row_lo;   // (the mask test below: chunks that end at or before the group's first row need no causal mask)
    const int frag_lane_off = (hh << 10) + (ql << 4);
    v8i qf[QPRE ? KS : 1];
    if (QPRE) {
#pragma unroll
        for (int s = 0; s < KS; s++) qf[s] = qfrag(s);
    }
    const int n_r = CAUSAL ? min(p.nchunks, min(row_hi, p.Sq - 1) / 64 + 1) : p.nchunks;
    const int per = (n_r + NW - 1) / NW;
    const int t0 = wave * per, t1 = min(n_r, t0 + per);
    v16f o[MB];
#pragma unroll
    for (int m = 0; m < MB; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[m][r] = 0.0f;
    float m_run = -1.0e30f, l_run = 0.0f;
    // The wave's next K chunk travels by LDS-DMA into its own CH bytes of `smem` (the merge slots alias them later) while the
    // current chunk is exponentiated: a chunk then waits for one L2 round trip (its V fragments, under the softmax) instead of
    // two in a row (K before the first MFMA); in registers the same prefetch spilled.
    static_assert(NW * CH <= 4 * SLOT, "the K prefetch areas fit the merge slots' LDS");
    unsigned char* kpre = smem + wave * CH;
    auto dma_k = [&](int t) {
        const unsigned char* src = kg + (long)t * CH + (lane << 4);
#pragma unroll
        for (int i = 0; i < CH / 1024; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024),
                                             (__attribute__((address_space(3))) void*)(kpre + i * 1024), 16, 0, 0);
    };
    if (t0 < t1) dma_k(t0);
    for (int t = t0; t < t1; t++) {
        const unsigned char* kc = kpre + frag_lane_off;
        const unsigned char* vc = vg + (long)t * CH + frag_lane_off;
        v16f s0, s1;
#pragma unroll
        for (int r = 0; r < 16; r++) { s0[r] = 0.0f; s1[r] = 0.0f; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // K(t) is in LDS
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const v8i ka = lds_read_frag(kc + ((0 * KS + s) << 11)), kb = lds_read_frag(kc + ((1 * KS + s) << 11));
            const v8i qs = QPRE ? qf[QPRE ? s : 0] : qfrag(s);
            s0 = mfma_f8<QK_FMT, QK_FMT>(ka, qs, s0);
            s1 = mfma_f8<QK_FMT, QK_FMT>(kb, qs, s1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ... and in registers: the area can take K(t + 1)
        if (t + 1 < t1) dma_k(t + 1);
        v8i vf[VB];
#pragma unroll
        for (int m = 0; m < VB; m++) vf[m] = gload_frag(vc + (m << 11));   // in flight under the softmax
        const int vsx = vx ? vscale_word(vx[min(t, kVxWords - 1)]) : kScaleWordOne;   // (vx: raw bytes from global memory or ready words from LDS)
        prep_scores<CAUSAL, TOKEN>(s0, s1, p, t * 64, r0, row, hh, skt);
        float mx = fmaxf(fmaxf(s0[0], s0[1]), s0[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s0[r]), s0[r + 1]);
        mx = fmaxf(mx, s0[15]);
#pragma unroll
        for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, s1[r]), s1[r + 1]);
        {
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        if (__any((mx - m_run) * c > kRescaleThr)) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
#pragma unroll
            for (int m = 0; m < MB; m++)
#pragma unroll
                for (int r = 0; r < 16; r++) o[m][r] *= alpha;
            l_run *= alpha;
            m_run = m_new;
        }
        const float mc = kPShift - m_run * c;
        v8i ph, pl;
        float ls = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            const v16f& sx = w < 4 ? s0 : s1;
            const int j = w & 3;
            float e[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { e[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[4 * j + i], c, mc)); ls += e[i]; }
            int hi = cvt_pk_fp8<QATTN_FMT_E4M3, false>(e[0], e[1], 0);
            hi = cvt_pk_fp8<QATTN_FMT_E4M3, true>(e[2], e[3], hi);
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 h01 = __builtin_amdgcn_cvt_pk_f32_fp8(hi, false), h23 = __builtin_amdgcn_cvt_pk_f32_fp8(hi, true);
            int lo = cvt_pk_fp8<QATTN_FMT_E4M3, false>(e[0] - h01[0], e[1] - h01[1], 0);
            lo = cvt_pk_fp8<QATTN_FMT_E4M3, true>(e[2] - h23[0], e[3] - h23[1], lo);
            ph[w] = hi; pl[w] = lo;
        }
        l_run += ls;
#pragma unroll
        for (int m0 = 0; m0 < MB; m0 += VB) {
            if (m0 > 0) {
#pragma unroll
                for (int m = 0; m < VB; m++) vf[m] = gload_frag(vc + ((m0 + m) << 11));
            }
#pragma unroll
            for (int m = 0; m < VB; m++) {
                if constexpr (VSCALE) {   // (the D = 128 kernel's call: V may be block-scaled)
                    o[m0 + m] = mfma_pv<V_FMT, QATTN_FMT_E4M3, true>(vf[m], ph, o[m0 + m], vsx);
                    o[m0 + m] = mfma_pv<V_FMT, QATTN_FMT_E4M3, true>(vf[m], pl, o[m0 + m], vsx);
                } else {
                    o[m0 + m] = mfma_f8<V_FMT, QATTN_FMT_E4M3>(vf[m], ph, o[m0 + m]);
                    o[m0 + m] = mfma_f8<V_FMT, QATTN_FMT_E4M3>(vf[m], pl, o[m0 + m]);
                }
            }
        }
    }
    // ---- merge the NW partials pairwise through LDS: {4..7} -> {0..3}, {2,3} -> {0,1}, {1} -> {0}
    __syncthreads();   // every wave is done with its K prefetch area, which the slots alias
    float* slots = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int half = NW / 2; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
            float* d = slots + (wave - half) * (SLOT / 4);
#pragma unroll
            for (int m = 0; m < MB; m++)
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++)
                    *reinterpret_cast<v4f*>(d + ((m * 4 + r4) * 64 + lane) * 4) = v4f{o[m][4 * r4], o[m][4 * r4 + 1], o[m][4 * r4 + 2], o[m][4 * r4 + 3]};
            d[MB * 16 * 64 + lane] = m_run;
            d[(MB * 16 + 1) * 64 + lane] = l_run;
        }
        __syncthreads();
        if (wave < half) {
            const float* d = slots + wave * (SLOT / 4);
            const float m_b = d[MB * 16 * 64 + lane], l_b = d[(MB * 16 + 1) * 64 + lane];
            const float m_new = fmaxf(m_run, m_b);
            const float fa = __builtin_amdgcn_exp2f((m_run - m_new) * c), fb = __builtin_amdgcn_exp2f((m_b - m_new) * c);
#pragma unroll
            for (int m = 0; m < MB; m++)
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) {
                    const v4f ob = *reinterpret_cast<const v4f*>(d + ((m * 4 + r4) * 64 + lane) * 4);
#pragma unroll
                    for (int i = 0; i < 4; i++) o[m][4 * r4 + i] = o[m][4 * r4 + i] * fa + ob[i] * fb;
                }
            l_run = l_run * fa + l_b * fb;
            m_run = m_new;
        }
        __syncthreads();
    }
    if (wave == 0) {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
        const float l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        const float sv = p.sv ? p.sv[kv_head] : 1.0f;
        store_o_rows<MB>(p.out, p.out_fmt, o, sv / l_tot, out_row_offset(p, bh, row, MB * 64), hh, store && row < p.Sq);
        if (p.lse && hh == 0 && store && row < p.Sq)
            p.lse[bh * p.lse_stride + row] = (0.6931471805599453f * (m_run * c - kPShift) + __logf(l_tot)) * p.lse_mul;
    }
}

template <int D, int NW, int QK_FMT, int V_FMT, bool CAUSAL, bool TOKEN, bool QLDS = false, bool VSCALE = false, typename QFrag>
__device__ __forceinline__ void rescue_rows(const AttnParams& p, unsigned char* smem, const unsigned char* kg, const unsigned char* vg,
                                            int r0, int wave, int lane, long bh, long kv_head, float c, const float* skt, QFrag&& qfrag,
                                            const unsigned* vx = nullptr) {
    rescue_rows_at<D, NW, QK_FMT, V_FMT, CAUSAL, TOKEN, QLDS, VSCALE>(p, smem, kg, vg, r0 + (lane & 31), true, r0, r0 + kQPerWave - 1, wave, lane, bh, kv_head, c,
                                                                     skt, qfrag, vx);
}

// A second stream for launches of one call that do not depend on each other (qattn_api.hip): the early rows of a causal call -- a few
// hundred short, latency-bound workgroups -- beside the main launch.  fork: `side` waits for everything on `st` so far; join: `st` waits
// for everything on `side`.  One stream and two events per host thread and device, created on first use outside a stream capture; nullptr
// (run the launches one after the other on `st`) while capturing before that first use, or when creation fails.  Inside a capture the
// fork / join pair becomes two parallel branches of the graph.
hipStream_t side_stream_fork(hipStream_t st);
int side_stream_join(hipStream_t st, hipStream_t side);

// kernel-file entry points (one translation unit per operand format / head dimension, see build.py)
int launch_attn_v2_e4m3(const AttnParams& p, int causal, int scale_mode, hipStream_t st);
int launch_attn_v2_e5m2(const AttnParams& p, int causal, int scale_mode, hipStream_t st);
int launch_attn_v2_e4m3_sv(const AttnParams& p, int causal, int scale_mode, hipStream_t st);   // the same kernels addressing q16 / v16 / out through strides
int launch_attn_v2_e5m2_sv(const AttnParams& p, int causal, int scale_mode, hipStream_t st);
int launch_attn_v4_d64(const AttnParams& p, int fmt, int causal, int scale_mode, hipStream_t st);
int launch_attn_v4_d128(const AttnParams& p, int fmt, int causal, int scale_mode, hipStream_t st);
int launch_attn_v4_d256(const AttnParams& p, int fmt, int causal, int scale_mode, hipStream_t st);

// true when the hand-scheduled kernel (qattn_attn_v2.hip) covers the case: D = 128 with head-wise scales.  Token-wise scales
// need 32 more registers per chunk for the per-key factors, which does not fit 256 registers at two waves per SIMD next
// to the two-term pass: those calls run on the templated kernel (qattn_attn_v4.hip).
// true when the 16-bit Q / V the kernels read and the output are dense [B,H,S,D] (the dense units of the twice-built kernels: QATTN_STRIDED16)
inline bool attn_params_dense(const AttnParams& p, int D) {
    const long rb = 2L * D;
    return (!p.q16 || (p.q16_rs == rb && p.q16_hs == rb * p.Sq && p.q16_bs == rb * p.Sq * p.Hq)) &&
           (!p.v16 || (p.v16_rs == rb && p.v16_hs == rb * p.Skv && p.v16_bs == rb * p.Skv * p.Hkv)) &&
           p.o_rs == rb && p.o_hs == rb * p.Sq && p.o_bs == rb * p.Sq * p.Hq;
}
inline bool attn_v2_covers(int D, int causal, int scale_mode) {
    (void)causal;
    return D == 128 && scale_mode == QATTN_SCALE_HEAD;
}
inline int launch_attn_v2(const AttnParams& p, int D, int fmt, int causal, int scale_mode, hipStream_t st) {
    if (!attn_v2_covers(D, causal, scale_mode)) return QATTN_ERR_UNSUPPORTED_DIM;
    // dense tensors (every call but a fused one on views): the instantiations with compile-time row sizes (QATTN_STRIDED16 above)
    if (!attn_params_dense(p, D)) return fmt == QATTN_FMT_E4M3 ? launch_attn_v2_e4m3_sv(p, causal, scale_mode, st) : launch_attn_v2_e5m2_sv(p, causal, scale_mode, st);
    return fmt == QATTN_FMT_E4M3 ? launch_attn_v2_e4m3(p, causal, scale_mode, st) : launch_attn_v2_e5m2(p, causal, scale_mode, st);
}
inline int launch_attn_v4_full(const AttnParams& p, int D, int fmt, int causal, int scale_mode, hipStream_t st) {
    if (D == 64) return launch_attn_v4_d64(p, fmt, causal, scale_mode, st);
    if (D == 128) return launch_attn_v4_d128(p, fmt, causal, scale_mode, st);
    if (D == 256) return launch_attn_v4_d256(p, fmt, causal, scale_mode, st);
    return QATTN_ERR_UNSUPPORTED_DIM;
}


}  // namespace qattn
