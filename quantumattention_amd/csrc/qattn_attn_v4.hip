// qattn_attn_v4.hip -- FP8 fused attention forward, "three waves per SIMD" structure (D = 128, byte-exponential path).
//
// Same math, operand layouts and C-ABI entry as qattn_attn_v2.hip (replaces fwd_attend_ker, tk/attention.py:97-349), but a
// different way of keeping the matrix pipe busy.  v2 software-pipelines QK^T / softmax / PV inside each wave and needs
// ~222 registers, so a SIMD holds two waves whose non-MFMA phases (checks, staging, LDS waits) are aligned by the
// workgroup barrier and leave the pipe idle ~30 % of the time.  Here every wave runs the plain sequence
//     QK^T(t) -> max / rescale -> exponentials -> PV(t)
// with single-buffered S and P (<= 168 registers), a workgroup is 4 waves x 32 rows with a two-stage K/V ring
// (48 KiB LDS), and a CU holds THREE independent workgroups: each SIMD interleaves three waves that belong to different
// workgroups, hence are never phase-aligned, and one wave's softmax hides under the other two's MFMAs.
// Templated on the head dimension: the whole forward for D = 64 / 256 and the D = 128 cases the hand-scheduled kernel does
// not cover (token-wise scales).  Precision modes: the one-term launch records, per (head, 32-row
// group = one wave's rows), whether a row ended peaked (R = l / p_max < peak_r0) in p.flags.  Two more launches follow:
// rescue_groups_kernel recomputes the flagged groups of every 256-row block that has at most kMaxRescueWaves of them
// (split-K over 8 waves, as inside the D = 128 kernel); the exact two-term variant of this kernel redoes the blocks with
// more.  Workgroups with nothing to do return at once.
#include "qattn_attn.h"
#include "qattn_pv16.h"

namespace qattn {

constexpr int kStages4 = 2;
// waves per workgroup: 4 (128 query rows; three workgroups per CU) for D <= 128, 8 for D = 256 (one workgroup per CU:
// 64 + 64 KiB of LDS, O^T alone is 128 registers, so two waves per SIMD)
// LIGHT = head-wise byte-exponential kernel (the lean register budget); token-wise / exact variants get one wave less.
template <int D, bool LIGHT> struct V4Shape {
    static constexpr int NW = D == 256 ? 8 : 4;
    static constexpr int WPS = D == 256 ? 2 : (D == 64 ? (LIGHT ? 4 : 3) : (LIGHT ? 3 : 2));
};

// 4 scores -> 4 e4m3 bytes of 2^x (see byte_group in qattn_attn_v2.hip: fma x4, v_cvt_pknorm_u16_f32 x2, v_perm_b32)
__device__ __forceinline__ int byte_exp4(float s0, float s1, float s2, float s3, float c8, float off8) {
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    // (v_pk_fma_f32 for the two pairs: 4-6 % SLOWER even at D = 64, where this kernel is VALU-bound -- profiles/r03/ab_pkfma_d64.log)
    const us2 qa = __builtin_amdgcn_cvt_pknorm_u16(__builtin_fmaf(s0, c8, off8), __builtin_fmaf(s1, c8, off8));
    const us2 qb = __builtin_amdgcn_cvt_pknorm_u16(__builtin_fmaf(s2, c8, off8), __builtin_fmaf(s3, c8, off8));
    unsigned ua, ub;
    __builtin_memcpy(&ua, &qa, 4);
    __builtin_memcpy(&ub, &qb, 4);
    return (int)__builtin_amdgcn_perm(ub, ua, 0x06040200u);
}

// BYTE: byte-exponential P + matrix-pipe row sums (default).  !BYTE: exact v_exp_f32, RNE fp8 conversion, fp32 row sums
// (LSE output), and -- when `two` is set for the launch -- the hi+lo two-term P for rows that see few keys.
template <int D, int QK_FMT, int V_FMT, bool CAUSAL, bool TOKEN, bool BYTE>
__global__ __launch_bounds__((V4Shape<D, (BYTE && !TOKEN)>::NW * 64), (V4Shape<D, (BYTE && !TOKEN)>::WPS))
void attn_fwd_kernel_v4(const AttnParams p, const int qb_lo, const int qb_n, const int mode) {
    const int two = mode & 1;            // exact kernels: hi + lo (two-term) P
    const bool only_flagged = mode & 2;  // redo launch: only the 256-row blocks with more than kMaxRescueWaves flagged groups
    const int redo_above = (mode & 4) ? 0 : kMaxRescueWaves;  // (mode & 4: no rescue launch ran, every flagged block is redone)
    constexpr int NW = V4Shape<D, (BYTE && !TOKEN)>::NW, kQPerWG4 = NW * kQPerWave;
    constexpr int CH = 64 * D, STAGE = 2 * CH + (TOKEN ? 256 : 0), MB = D / 32, KS = D / 64;   // token-wise: + the chunk's 64 key scales
    constexpr int RK = CH / (NW * 1024);   // 1 KiB DMA pieces per wave for the K (and for the V) part of a stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hh = lane >> 5;

    int head, qb;
    map_block(p, blockIdx.x, qb_n, CAUSAL, head, qb);
    qb += qb_lo;
    const int b = head / p.Hq, h = head % p.Hq;
    const long kv_head = (long)b * p.Hkv + h / (p.Hq / p.Hkv);
    const int q0_wg = qb * kQPerWG4, q0 = q0_wg + wave * kQPerWave, qrow = q0 + ql;
    const long bh = (long)b * p.Hq + h;
    const int ng = (p.Sq + 31) >> 5;          // 32-row groups per head
    unsigned* flag = p.flags ? p.flags + bh * ng + (q0 >> 5) : nullptr;   // this wave's group
    if (mode == 0 && flag != nullptr && p.ssq_q != nullptr) {
        // AUTO in the fused step: a head whose predicted score spread makes its rows end below kPeakR0 anyway (predicted_r,
        // qattn_attn.h) is not swept with one-term P first: its groups are flagged, the redo launch attends them in two-term mode
        const int kvh = (int)kv_head;
        float var = sum_partials(p.ssq_q + bh * p.ssq_stride, p.ssq_n, lane) * sum_partials(p.ssq_k + (long)kvh * p.ssq_stride, p.ssq_n, lane) * p.var_mul;
        if (var >= kVarDeadband) {
            const int nkeys = CAUSAL ? min(p.Skv, q0_wg + 1) : p.Skv;
            if (__builtin_amdgcn_readfirstlane(predicted_r((float)nkeys, var, fmaxf(p.peak_z, kPeakZWide)) < kPeakR0 ? 1 : 0)) {
                if (lane == 0 && q0 < p.Sq) *flag = 1u;
                return;
            }
        }
    }
    if (only_flagged) {  // workgroup-uniform: count the flagged groups of the 256-row block these rows belong to
        const int g0 = (q0_wg >> 8) << 3;
        int nf = 0;
        for (int g = g0; g < min(g0 + 8, ng); g++) nf += p.flags[bh * ng + g] != 0u;
        if (nf <= redo_above) return;
    }
    const unsigned char* kg_w = p.k + kv_head * (long)p.nchunks * CH + (wave << 10);
    const unsigned char* vg_w = p.v + kv_head * (long)p.nchunks * CH + (wave << 10);
    const int n_wg = CAUSAL ? min(p.nchunks, (min(q0_wg + kQPerWG4, p.Sq) - 1) / 64 + 1) : p.nchunks;
    const int n_w = CAUSAL ? min(n_wg, (q0 + kQPerWave - 1) / 64 + 1) : p.nchunks;

    const float* skt = TOKEN ? p.sk + kv_head * p.Skv : nullptr;
    // stage(t) = {K chunk t, V chunk t} -> slot t & 1; every wave copies 2*RK x 1 KiB of it by LDS-DMA
    const unsigned lane16 = (unsigned)lane << 4;
    unsigned coff = 0, slot_next = 0;
    auto dma_next = [&]() {
        unsigned char* dst = smem + slot_next + (wave << 10);
#pragma unroll
        for (int r = 0; r < RK; r++) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kg_w + (coff + lane16 + r * (NW * 1024))),
                                             (__attribute__((address_space(3))) void*)(dst + r * (NW * 1024)), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vg_w + (coff + lane16 + r * (NW * 1024))),
                                             (__attribute__((address_space(3))) void*)(dst + CH + r * (NW * 1024)), 16, 0, 0);
        }
        if (TOKEN && wave == 0) {
            // the chunk's 64 per-key scales ride along with the stage (one 4-byte LDS-DMA piece per lane): read from global
            // memory right before their use they exposed a load latency per chunk (1.20 ms at the C2 shape)
            const int key = min((int)(coff / CH) * 64 + lane, p.Skv - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(skt + key),
                                             (__attribute__((address_space(3))) void*)(smem + slot_next + 2 * CH), 4, 0, 0);
        }
        coff += CH;
        slot_next ^= STAGE;
    };
    dma_next();

    // Q^T fragments parked in this lane's own LDS slots (registers are the scarce resource at three waves per SIMD)
    unsigned char* qbuf = smem + kStages4 * STAGE + wave * (KS << 11) + (hh << 10) + (ql << 4);
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
    float c;
    if (TOKEN) c = p.sm_log2e * (qrow < p.Sq ? p.sq[((long)b * p.Hq + h) * p.Sq + qrow] : 1.0f);
    else c = p.sm_log2e * p.sq[(long)b * p.Hq + h] * p.sk[kv_head];

    v16f o[MB];
#pragma unroll
    for (int m = 0; m < MB; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[m][r] = 0.0f;
    v4f lsum = {0.0f, 0.0f, 0.0f, 0.0f};
    v4f lsq = {0.0f, 0.0f, 0.0f, 0.0f};   // one-term AUTO launch: row sums of P'^2 (WaveState::lsq in qattn_attn_v2.hip)
    float l2_run = 0.0f;
    const bool neff = mode == 0 && p.peak_neff > 0.0f;   // launch-uniform
    v8i ones;  // A of the row-sum MFMA (see WaveState::lsum in qattn_attn_v2.hip)
    {
        const int row = lane & 15, kg = lane >> 4;
        const int one = ((row == 0 && !(kg & 1)) || (row == 1 && (kg & 1))) ? 0x38383838 : 0;
#pragma unroll
        for (int w = 0; w < 8; w++) ones[w] = one;
    }
    float m_run = -1.0e30f, m_true = -1.0e30f, l_run = 0.0f;
    float lim = -1.0e30f, off8 = 0.0f;   // m_run + thr / c and the byte formula's additive constant (set by the first chunk's fix-up)
    constexpr float U16 = 1.0f / 65535.0f;
    const float c8 = (8.0f * U16) * c;
    const int frag_lane_off = (hh << 10) + (ql << 4);

    // AUTO, one-term launch: first-chunk score-variance forecast (see kv_sweep in qattn_attn_v2.hip): every wave votes after
    // chunk 0; if the workgroup's share of a 256-row block is going to be flagged anyway, its groups are flagged now and the
    // two-term redo launch attends them -- 2 of n chunks spent instead of all.
    unsigned* vote = reinterpret_cast<unsigned*>(smem + kStages4 * STAGE + NW * kQPerWave * D);
    // block-scaled V (fused step, head-wise scales: the pre-pass gives every 64-key chunk of V its own power-of-two scale, one
    // E8M0 byte per chunk in p.vexp): the head's bytes are kept in LDS behind the votes -- the loop's first barrier publishes
    // them -- and ride into the PV products as the MFMA's A scale.  127 = 2^0 where V has one scale per head.
    constexpr bool VS = !TOKEN;
    unsigned* vxl = vote + 16;
    if (VS) {
        for (int i = tid; i < kVxWords; i += NW * 64) vxl[i] = (unsigned)vscale_word((p.vexp && i < p.nchunks) ? p.vexp[kv_head * p.vexp_stride + i] : 127u);
    }
    const int vx_step = (VS && p.vexp != nullptr) ? 1 : 0;   // (no table entries beyond nchunks <= kVxWords; without vexp entry 0 = 2^0)
    const bool forecast = mode == 0 && flag != nullptr && p.peak_r0 > 0.0f && n_wg >= 16 && (!CAUSAL || q0_wg >= 64);
    for (int t = 0; t < n_wg; t++) {
        // Q^T fragments do not depend on the stage: request them before the barrier so their LDS latency hides behind it
        v8i qf[KS];
#pragma unroll
        for (int s = 0; s < KS; s++) qf[s] = lds_read_frag(qbuf + (s << 11));
        wait_vmcnt<0>();                  // this wave's pieces of stage t have landed
        __builtin_amdgcn_s_barrier();     // ... and everyone's; every wave is also done with stage t-1's slot
        if (forecast && t == 1) {   // workgroup-uniform
            int nf = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) nf += vote[w] != 0u ? 1 : 0;
            if (__builtin_amdgcn_readfirstlane(nf) * 8 > kMaxRescueWaves * NW) {
                if (lane == 0 && q0 < p.Sq) *flag = 1u;
                return;   // (no LDS-DMA of this wave is in flight: stage 1 was waited for above, stage 2 is not requested yet)
            }
        }
        if (t + 1 < n_wg) dma_next();
        if (t >= n_w) continue;           // causal: this wave's rows end before chunk t (it keeps the barrier / DMA cadence)
        const int vsx = VS ? (int)vxl[t * vx_step] : kScaleWordOne;   // this chunk's V scale byte (lands under the QK^T MFMAs and the softmax)
        const unsigned char* kbuf = smem + (t & 1) * STAGE + frag_lane_off;
        const unsigned char* vbuf = kbuf + CH;
        // ---- S^T = K.Q^T
        v16f s0, s1;
#pragma unroll
        for (int r = 0; r < 16; r++) { s0[r] = 0.0f; s1[r] = 0.0f; }
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const v8i ka = lds_read_frag(kbuf + ((0 * KS + s) << 11)), kb = lds_read_frag(kbuf + ((1 * KS + s) << 11));
            s0 = mfma_f8<QK_FMT, QK_FMT>(ka, qf[s], s0);
            s1 = mfma_f8<QK_FMT, QK_FMT>(kb, qf[s], s1);
        }
        // the first two V fragments travel while the softmax runs (all four would not fit in 168 registers)
        const v8i vf0 = lds_read_frag(vbuf), vf1 = lds_read_frag(vbuf + (1 << 11));
        // ---- token-wise key scales, ragged-tail / causal mask (rare or cheap)
        const int k0 = t * 64;
        if (TOKEN) {
#pragma unroll
            for (int tt = 0; tt < 2; tt++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float4 w = *reinterpret_cast<const float4*>(kbuf - frag_lane_off + 2 * CH + (32 * tt + 8 * j + 4 * hh) * 4);
                    v16f& sx = tt ? s1 : s0;
                    sx[4 * j + 0] *= w.x; sx[4 * j + 1] *= w.y; sx[4 * j + 2] *= w.z; sx[4 * j + 3] *= w.w;
                }
        }
        const bool need_mask = (k0 + 64 > p.Skv) || (CAUSAL && k0 + 63 > q0);   // (a single compare `t == n_w - 1` measured no faster: D = 64 +-0, D = 256 +1.7 %)
        if (__builtin_expect(need_mask, 0)) {
#pragma unroll
            for (int r = 0; r < 32; r++) {
                const int key = k0 + 32 * (r >> 4) + (r & 3) + 8 * ((r & 15) >> 2) + 4 * hh;
                const bool dead = key >= p.Skv || (CAUSAL && key > qrow);
                v16f& sx = (r >> 4) ? s1 : s0;
                sx[r & 15] = dead ? -INFINITY : sx[r & 15];
            }
        }
        if (forecast && t == 0) {
            float su = 0.0f, sq2 = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                su += s0[r] + s1[r];
                sq2 = __builtin_fmaf(s0[r], s0[r], __builtin_fmaf(s1[r], s1[r], sq2));
            }
            su *= c; sq2 *= c * c;   // log2-domain scores (token-wise: c carries the row's scale)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { su += __shfl_xor(su, off); sq2 += __shfl_xor(sq2, off); }
            const float mean = su * (1.0f / 2048.0f), ln2 = 0.6931471805599453f;
            const float var = fmaxf(sq2 * (1.0f / 2048.0f) - mean * mean, 0.0f) * ln2 * ln2;
            const int nkeys = CAUSAL ? min(p.Skv, q0 + kQPerWave) : p.Skv;
            const bool mine = var >= kVarDeadband && predicted_r((float)nkeys, var, kPeakZWide) < kPeakR0;
            if (lane == 0) vote[wave] = mine ? 1u : 0u;
        }
        // ---- running max; rescale only when a row's max grew past the headroom of the shifted exponent
        float mx = max32_after_mfma(s0, s1);
        {
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            m_true = max3_raw(m_true, __uint_as_float(sw[0]), __uint_as_float(sw[1]));
            mx = max3_raw(__uint_as_float(sw[0]), __uint_as_float(sw[1]), __uint_as_float(sw[1]));
        }
        if (__builtin_expect(__any(mx > lim) != 0, 0)) {   // (mx - m_run) c > thr: P' could overflow e4m3
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
#pragma unroll
            for (int m = 0; m < MB; m++)
#pragma unroll
                for (int r = 0; r < 16; r++) o[m][r] *= alpha;
            if (BYTE) {
                const float alpha16 = __shfl(alpha, (lane & 15) + 16);
                lsum[0] *= alpha;
                lsum[1] *= alpha16;
                lsq[0] *= alpha * alpha;
                lsq[1] *= alpha16 * alpha16;
            } else {
                l_run *= alpha;
                l2_run *= alpha * alpha;
            }
            m_run = m_new;
            // the two values that change only here, kept instead of re-derived every chunk
            lim = m_new + kRescaleThrByte / c;
            off8 = __builtin_fmaf((-8.0f * U16) * m_new, c, (8.0f * kPShiftByte + 56.0f + kByteBias) * U16);
        }
        v8i pv, pl;
        if (BYTE) {
            // ---- P = e4m3 bytes of 2^(c*(s - m) + shift)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                pv[j] = byte_exp4(s0[4 * j], s0[4 * j + 1], s0[4 * j + 2], s0[4 * j + 3], c8, off8);
                pv[4 + j] = byte_exp4(s1[4 * j], s1[4 * j + 1], s1[4 * j + 2], s1[4 * j + 3], c8, off8);
            }
        } else {
            // ---- exact exponentials, RNE e4m3; optional residual term lo = fp8(p - hi); fp32 row sums
            const float mc = kPShift - m_run * c;
            float ls = 0.0f, ls2 = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; w++) {
                const v16f& sx = w < 4 ? s0 : s1;
                const int j = w & 3;
                float e[4];
#pragma unroll
                for (int i = 0; i < 4; i++) { e[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[4 * j + i], c, mc)); ls += e[i]; }
                if (neff) {
#pragma unroll
                    for (int i = 0; i < 4; i++) ls2 = __builtin_fmaf(e[i], e[i], ls2);
                }
                int ph = cvt_pk_fp8<QATTN_FMT_E4M3, false>(e[0], e[1], 0);
                ph = cvt_pk_fp8<QATTN_FMT_E4M3, true>(e[2], e[3], ph);
                pv[w] = ph;
                int plo = 0;
                if (two) {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    const f2 h01 = __builtin_amdgcn_cvt_pk_f32_fp8(ph, false), h23 = __builtin_amdgcn_cvt_pk_f32_fp8(ph, true);
                    plo = cvt_pk_fp8<QATTN_FMT_E4M3, false>(e[0] - h01[0], e[1] - h01[1], 0);
                    plo = cvt_pk_fp8<QATTN_FMT_E4M3, true>(e[2] - h23[0], e[3] - h23[1], plo);
                }
                pl[w] = plo;
            }
            l_run += ls;
            l2_run += ls2;
        }
        // ---- O^T += V^T.P^T, row sums
        o[0] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(vf0, pv, o[0], vsx);
        o[1] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(vf1, pv, o[1], vsx);
        if (!BYTE && two) {
            o[0] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(vf0, pl, o[0], vsx);
            o[1] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(vf1, pl, o[1], vsx);
        }
#pragma unroll
        for (int m = 2; m < MB; m += 2) {  // the remaining fragments land under the MFMAs already issued
            const v8i va = lds_read_frag(vbuf + (m << 11)), vb = lds_read_frag(vbuf + ((m + 1) << 11));
            o[m] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(va, pv, o[m], vsx);
            o[m + 1] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(vb, pv, o[m + 1], vsx);
            if (!BYTE && two) {
                o[m] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(va, pl, o[m], vsx);
                o[m + 1] = mfma_pv<V_FMT, QATTN_FMT_E4M3, VS>(vb, pl, o[m + 1], vsx);
            }
        }
        if (BYTE) lsum = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, pv, lsum, QATTN_FMT_E4M3, QATTN_FMT_E4M3, 0, 0, 0, 0);
        // (the same bytes read as e5m2 ~= P'^2 / 2)
        if (BYTE && neff) lsq = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, pv, lsq, QATTN_FMT_E4M3, QATTN_FMT_E5M2, 0, 0, 0, 0);
    }

    // ---- epilogue
    float l_tot, l2_tot;
    if (BYTE) {
        const float s0l = __shfl(lsum[0], lane & 15), s1l = __shfl(lsum[1], lane & 15);
        l_tot = (lane & 16) ? s1l : s0l;
        const float t0l = __shfl(lsq[0], lane & 15), t1l = __shfl(lsq[1], lane & 15);
        l2_tot = (lane & 16) ? t1l : t0l;
    } else {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
        l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(l2_run), __float_as_uint(l2_run), false, false);
        l2_tot = __uint_as_float(sw2[0]) + __uint_as_float(sw2[1]);
    }
    constexpr float SHIFT = BYTE ? kPShiftByte : kPShift;
    if (p.peak_r0 > 0.0f && !two && !only_flagged) {
        // one-term launch: a peaked row (row_is_peaked, qattn_attn.h: R = l' / p'_max and the effective key count) flags its group
        const float r_inv_pmax = __builtin_amdgcn_exp2f(-(SHIFT + (m_true - m_run) * c));
        const float nkeys_row = (float)(CAUSAL ? min(qrow + 1, p.Skv) : p.Skv);
        const bool peaked = qrow < p.Sq && (neff ? row_is_peaked<BYTE, true>(p, l_tot, l2_tot, r_inv_pmax, m_true == m_run, nkeys_row)
                                                 : row_is_peaked<BYTE, false>(p, l_tot, l2_tot, r_inv_pmax, false, nkeys_row));
        if (__any(peaked) && lane == 0 && q0 < p.Sq) *flag = 1u;
    }
    const float sv = p.sv ? p.sv[kv_head] : 1.0f;
    const float inv = sv / l_tot;
    store_o_rows<MB>(p.out, p.out_fmt, o, inv, out_row_offset(p, bh, qrow, MB * 64), hh, qrow < p.Sq);
    if (!BYTE && p.lse && hh == 0 && qrow < p.Sq)  // ln sum_j exp(score_j) = ln2 * (m*c - shift) + ln(l')
        p.lse[bh * p.lse_stride + qrow] = (0.6931471805599453f * (m_run * c - kPShift) + __logf(l_tot)) * p.lse_mul;
    if (!BYTE && two && p.path && hh == 0 && qrow < p.Sq) p.path[bh * p.Sq + qrow] = (unsigned char)QATTN_PATH_TWO_TERM;   // (fused entry's debug output)
}

template <int D, int FMT, bool CAUSAL, bool TOKEN, bool BYTE>
static int launch_v4_one(const AttnParams& p, int row_lo, int row_hi, int mode, hipStream_t st) {
    constexpr int NW = V4Shape<D, (BYTE && !TOKEN)>::NW, ROWS = NW * kQPerWave;
    const int qb_lo = row_lo / ROWS, qb_n = ceil_div(min(row_hi, p.Sq), ROWS) - qb_lo;
    if (qb_n <= 0) return QATTN_OK;
    const int grid = p.B * p.Hq * qb_n;
    const size_t lds = (size_t)kStages4 * (2 * 64 * D + (TOKEN ? 256 : 0)) + (size_t)NW * kQPerWave * D + 64 + (TOKEN ? 0 : 4 * kVxWords);  // K/V ring (+ key scales) + parked Q^T fragments + forecast votes + V chunk scale bytes
    auto kern = attn_fwd_kernel_v4<D, FMT, FMT, CAUSAL, TOKEN, BYTE>;
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return QATTN_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, p, qb_lo, qb_n, mode);
    return QATTN_OK;
}

template <int D, bool BYTE>
static int launch_v4_d(const AttnParams& p, int fmt, int causal, int scale_mode, int row_lo, int row_hi, int mode, hipStream_t st) {
    const bool tok = scale_mode == QATTN_SCALE_TOKEN;
#define QATTN_V4(F, C, T) return launch_v4_one<D, F, C, T, BYTE>(p, row_lo, row_hi, mode, st)
    if (fmt == QATTN_FMT_E4M3) {
        if (causal) { if (tok) QATTN_V4(QATTN_FMT_E4M3, true, true); else QATTN_V4(QATTN_FMT_E4M3, true, false); }
        else { if (tok) QATTN_V4(QATTN_FMT_E4M3, false, true); else QATTN_V4(QATTN_FMT_E4M3, false, false); }
    } else {
        if (causal) { if (tok) QATTN_V4(QATTN_FMT_E5M2, true, true); else QATTN_V4(QATTN_FMT_E5M2, true, false); }
        else { if (tok) QATTN_V4(QATTN_FMT_E5M2, false, true); else QATTN_V4(QATTN_FMT_E5M2, false, false); }
    }
#undef QATTN_V4
}

// Rescue launch: one 8-wave workgroup per 256-row block; the flagged 32-row groups of a block with at most kMaxRescueWaves of
// them are recomputed by rescue_rows (qattn_attn.h), the Q^T fragments fetched from the row-major q8 tensor.
// 256-row blocks a workgroup of the rescue launch looks at (64 flag words: one per lane).  Non-causal calls only: there the launch is a few
// dozen rescues among thousands of empty workgroups; a causal call flags a quarter of its blocks (the rows that see 1 .. 2 k keys), and
// several rescues in a row per workgroup made its launch 20 .. 70 % longer (profiles/r05/ab_v4_rescue_scan_*).
// (D = 256: one block, too -- the loop around rescue_rows costs that instantiation three spilled registers)
template <bool CAUSAL, int D> constexpr int rescue_scan() { return (CAUSAL || D == 256) ? 1 : 8; }
template <int D, int QK_FMT, int V_FMT, bool CAUSAL, bool TOKEN>
__global__ __launch_bounds__(512, 2) void rescue_groups_kernel(const AttnParams p, const int blk_lo, const int blk_n) {
    constexpr int CH = 64 * D;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ql = lane & 31, hh = lane >> 5;
    // One workgroup looks at rescue_scan() 256-row blocks (of the (b, h, block) order: 8 flag words each, one word per lane, ONE
    // memory round trip) and recomputes what it finds.  One workgroup per block -- rounds 3 and 4 -- made the launch as long as its
    // thousands of empty workgroups' round trips: 65 us at B16 H16 S8192 with a few dozen groups to recompute (dev kernel trace,
    // profiles/r05/kstats_d64_reference_shape.txt).
    const int ng = (p.Sq + 31) >> 5;
    const int nb = p.B * p.Hq * blk_n;
    // (blocks blockIdx.x, + gridDim.x, ...: the flagged blocks of a causal call are the same few of every head, side by side in that order)
    const int n0 = (int)blockIdx.x + (lane >> 3) * (int)gridDim.x, g_l = lane & 7;
    unsigned long long found;
    {
        const int bh_l = n0 / blk_n, blk_l = blk_lo + n0 % blk_n;
        const bool valid = (lane >> 3) < rescue_scan<CAUSAL, D>() && n0 < nb && blk_l * 8 + g_l < ng;
        const unsigned f = valid ? p.flags[(long)bh_l * ng + blk_l * 8 + g_l] : 0u;
        found = __ballot(f != 0u);   // byte i: the flagged groups of block blockIdx.x + i gridDim.x (the same in every wave)
    }
    for (int i = 0; i < rescue_scan<CAUSAL, D>(); i++) {
        const unsigned flagged = __builtin_amdgcn_readfirstlane((unsigned)(found >> (8 * i)) & 0xffu);
        if (flagged == 0u || __builtin_popcount(flagged) > kMaxRescueWaves) continue;   // nothing to do / redone by the two-term launch
        // (workgroup-uniform values, kept in scalar registers: rescue_rows needs every vector register it can get)
        const int n = __builtin_amdgcn_readfirstlane((int)blockIdx.x + i * (int)gridDim.x);
        const int bh_i = __builtin_amdgcn_readfirstlane(n / blk_n);
        const long bh = bh_i;
        const int blk = __builtin_amdgcn_readfirstlane(blk_lo + n % blk_n);
        const int b = bh_i / p.Hq, h = bh_i % p.Hq;
        const long kv_head = __builtin_amdgcn_readfirstlane(b * p.Hkv + h / (p.Hq / p.Hkv));
        const unsigned char* kg = p.k + kv_head * (long)p.nchunks * CH;
        const unsigned char* vg = p.v + kv_head * (long)p.nchunks * CH;
        const float* skt = TOKEN ? p.sk + kv_head * p.Skv : nullptr;
        for (unsigned rest = flagged; rest != 0u; rest &= rest - 1u) {
            const int r0 = blk * 256 + __builtin_ctz(rest) * kQPerWave, row = r0 + ql;
            const bool qvalid = row < p.Sq;
            float c;
            if (TOKEN) c = p.sm_log2e * (qvalid ? p.sq[bh * p.Sq + row] : 1.0f);
            else c = p.sm_log2e * p.sq[bh] * p.sk[kv_head];
            const unsigned char* qp = p.q + ((bh * p.Sq + (qvalid ? row : 0)) * D) + hh * 32;
            auto qfrag = [&](int s_) {
                v4i lo = *reinterpret_cast<const v4i*>(qp + s_ * 64);
                v4i hi = *reinterpret_cast<const v4i*>(qp + s_ * 64 + 16);
                if (!qvalid) { lo = v4i{0, 0, 0, 0}; hi = v4i{0, 0, 0, 0}; }
                return v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            };
            // (block-scaled V: the chunk scale bytes straight from global memory, one word per chunk)
            rescue_rows<D, 8, QK_FMT, V_FMT, CAUSAL, TOKEN, false, !TOKEN>(p, smem, kg, vg, r0, wave, lane, bh, kv_head, c, skt, qfrag,
                                                                           (!TOKEN && p.vexp) ? p.vexp + kv_head * p.vexp_stride : nullptr);
            // (fused entry's debug output; the row is re-derived from scalars + the lane index: nothing more is live across the call above)
            __syncthreads();   // (the next group's K prefetch areas alias this one's merge slots)
        }
    }
}

// The fused entry's row_path output for the rows the rescue launch recomputes (two-term fp8 P on the fp8 V): derived from the same flag
// words, by a launch of its own that exists only when the caller asked for the output -- a store inside rescue_groups_kernel cost its
// D = 256 instantiation (256 registers) a spill.  One thread per query row of the blocks from blk_lo on.
template <int D>   // (one instance per translation unit of this file)
__global__ void mark_rescued_rows_kernel(const unsigned* flags, unsigned char* path, int Sq, int blk_lo, int blk_n) {
    const int ng = (Sq + 31) >> 5;
    const long bh = blockIdx.x / blk_n;
    const int blk = blk_lo + (int)(blockIdx.x % blk_n), row = blk * 256 + (int)threadIdx.x;
    const unsigned* f = flags + bh * ng + blk * 8;
    int nf = 0;
    for (int g = 0; g < 8; g++) nf += (blk * 8 + g < ng && f[g] != 0u) ? 1 : 0;
    if (row < Sq && nf > 0 && nf <= kMaxRescueWaves && f[threadIdx.x >> 5] != 0u) path[bh * Sq + row] = (unsigned char)QATTN_PATH_TWO_TERM;
}

template <int D, int FMT, bool CAUSAL, bool TOKEN>
static int launch_rescue_one(const AttnParams& p, int row_lo, hipStream_t st) {
    const int blk_lo = row_lo / 256, blk_n = ceil_div(p.Sq, 256) - blk_lo;
    if (blk_n <= 0) return QATTN_OK;
    const size_t lds = 4 * (size_t)rescue_slot_bytes<D>();
    auto kern = rescue_groups_kernel<D, FMT, FMT, CAUSAL, TOKEN>;
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return QATTN_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(ceil_div(p.B * p.Hq * blk_n, rescue_scan<CAUSAL, D>())), dim3(512), lds, st, p, blk_lo, blk_n);
    return QATTN_OK;
}

template <int D>
static int launch_rescue_d(const AttnParams& p, int fmt, int causal, int scale_mode, int row_lo, hipStream_t st) {
    const bool tok = scale_mode == QATTN_SCALE_TOKEN;
#define QATTN_RS(F, C, T) return launch_rescue_one<D, F, C, T>(p, row_lo, st)
    if (fmt == QATTN_FMT_E4M3) {
        if (causal) { if (tok) QATTN_RS(QATTN_FMT_E4M3, true, true); else QATTN_RS(QATTN_FMT_E4M3, true, false); }
        else { if (tok) QATTN_RS(QATTN_FMT_E4M3, false, true); else QATTN_RS(QATTN_FMT_E4M3, false, false); }
    } else {
        if (causal) { if (tok) QATTN_RS(QATTN_FMT_E5M2, true, true); else QATTN_RS(QATTN_FMT_E5M2, true, false); }
        else { if (tok) QATTN_RS(QATTN_FMT_E5M2, false, true); else QATTN_RS(QATTN_FMT_E5M2, false, false); }
    }
#undef QATTN_RS
}

template <int D>
static int launch_rescue_head(const AttnParams& p, int fmt, int causal, int row_lo, hipStream_t st) {  // head-wise scales only
    if (fmt == QATTN_FMT_E4M3) return causal ? launch_rescue_one<D, QATTN_FMT_E4M3, true, false>(p, row_lo, st) : launch_rescue_one<D, QATTN_FMT_E4M3, false, false>(p, row_lo, st);
    return causal ? launch_rescue_one<D, QATTN_FMT_E5M2, true, false>(p, row_lo, st) : launch_rescue_one<D, QATTN_FMT_E5M2, false, false>(p, row_lo, st);
}

template <int D>
static int launch_v4_full_d(const AttnParams& pin, int fmt, int causal, int scale_mode, hipStream_t st) {
    AttnParams p = pin;
    p.tail_lo = 0;
    p.risky_lo = p.risky_hi = 0;   // (plain longest-first: these launches cover sub-ranges of a head's blocks, and their rescues are launches of their own)
    // leading rows (a multiple of 256) that run two-term P from the start: all of them (QATTN_PRECISION_ACCURATE) or those
    // that see fewer than kTwoTermKeys keys
    const int rows_all = ceil_div(p.Sq, 256) * 256;
    int rows_two;
    if (p.precision == QATTN_PRECISION_ACCURATE) rows_two = rows_all;
    // (the early set is pv16_early_blocks' -- the D = 128 kernel, the fp16 side launch, include/qattn.h and the tests' oracle all use it: with
    // fewer than two_term_keys keys EVERY block is early, causal or not, whatever Sq; ADVICE r4: clamped by the key count, causal calls
    // with Sq > Skv < 1024 left their later rows on one-term fp8 P)
    else rows_two = min(rows_all, pv16_early_blocks(p.Sq, p.Skv, causal, p.two_term_keys) * 256);
    const bool byte_exp = p.lse == nullptr;
    const bool rescue = p.peak_r0 > 0.0f && rows_two < p.Sq;
    if (rescue) {
        if (!p.flags) return QATTN_ERR_WORKSPACE;
        // (fused step: the quantise pass cleared them on its way, sched_zeroed)
        if (!p.sched_zeroed && zero_words(p.flags, (long)p.B * p.Hq * ceil_div(p.Sq, 32), st) != hipSuccess) return QATTN_ERR_LAUNCH;
    }
    int rc = QATTN_OK;
    // The rows that run two-term (or on the 16-bit V) from the start and the rest are disjoint pieces of the output with nothing
    // between them: where both exist -- a causal call -- the early rows go to a second stream, beside the main launch.  They are a few
    // hundred short workgroups whose chunks wait for one memory round trip each; alone on the chip they took 87 of the 297 us of a D = 64
    // causal AUTO call at B4 H32 S4096 (profiles/r04/trace_d64_causal_auto.txt).
    hipStream_t side = nullptr;
    if (rows_two > 0 && rows_two < p.Sq) side = side_stream_fork(st);
    hipStream_t st_e = side ? side : st;
    // (requested FIRST: their serial chains of memory round trips start at once, the main launch's workgroups fill the CUs around them)
    // the fused step (D = 64 / 256, and D = 128 with token-wise scales): the query blocks that see fewer than two_term_keys keys attend the ORIGINAL
    // 16-bit V with 16-bit P (qattn_pv16.h) instead of two-term fp8 P on the fp8 V; the rest of rows_two (ACCURATE) stays two-term
    int rows_early = 0;
    if (p.v16 != nullptr) {   // (with an LSE output too: pv16_block_pass writes its rows' entries)
        rows_early = min(rows_two, pv16_early_blocks(p.Sq, p.Skv, causal, p.two_term_keys) * 256);
        if (rc == QATTN_OK && rows_early > 0) rc = launch_attn_pv16(p, D, fmt, p.out_fmt, causal, scale_mode, st_e, rows_early / 256);
    }
    if (rc == QATTN_OK && rows_two > rows_early) rc = launch_v4_d<D, false>(p, fmt, causal, scale_mode, rows_early, rows_two, 1, st_e);
    if (rc == QATTN_OK && rows_two < p.Sq)
        rc = byte_exp ? launch_v4_d<D, true>(p, fmt, causal, scale_mode, rows_two, p.Sq, 0, st)
                      : launch_v4_d<D, false>(p, fmt, causal, scale_mode, rows_two, p.Sq, 0, st);
    // flagged 32-row groups: rescued one by one where a 256-row block has few of them, else the block is redone
    // (D = 256 with token-wise scales: the rescue loop does not fit 256 registers beside 128 of O^T -- every flagged block is redone)
    const bool group_rescue = !(D == 256 && scale_mode == QATTN_SCALE_TOKEN);
    if constexpr (D != 256) {
        if (rc == QATTN_OK && rescue) rc = launch_rescue_d<D>(p, fmt, causal, scale_mode, rows_two, st);
    } else {
        if (rc == QATTN_OK && rescue && group_rescue) rc = launch_rescue_head<D>(p, fmt, causal, rows_two, st);
    }
    if (rc == QATTN_OK && rescue && group_rescue && p.path != nullptr) {
        const int blk_lo = rows_two / 256, blk_n = ceil_div(p.Sq, 256) - blk_lo;
        if (blk_n > 0) hipLaunchKernelGGL(mark_rescued_rows_kernel<D>, dim3(p.B * p.Hq * blk_n), dim3(256), 0, st, p.flags, p.path, p.Sq, blk_lo, blk_n);
    }
    if (rc == QATTN_OK && rescue) rc = launch_v4_d<D, false>(p, fmt, causal, scale_mode, rows_two, p.Sq, group_rescue ? 3 : 7, st);
    if (side) {   // (joined on every path: a capture must not end with the side stream still forked)
        const int rj = side_stream_join(st, side);
        if (rc == QATTN_OK) rc = rj;
    }
    return rc;
}

// The whole forward on the templated kernel: D = 64 / 256, and D = 128 where qattn_attn_v2.hip does not apply.
// One translation unit per head dimension (build.py compiles this file three times with -DQATTN_ONLY_D=64|128|256, which keeps
// the build parallel); without the macro the file provides all three.
#if !defined(QATTN_ONLY_D) || QATTN_ONLY_D == 64
int launch_attn_v4_d64(const AttnParams& p, int fmt, int causal, int scale_mode, hipStream_t st) { return launch_v4_full_d<64>(p, fmt, causal, scale_mode, st); }
#endif
#if !defined(QATTN_ONLY_D) || QATTN_ONLY_D == 128
int launch_attn_v4_d128(const AttnParams& p, int fmt, int causal, int scale_mode, hipStream_t st) { return launch_v4_full_d<128>(p, fmt, causal, scale_mode, st); }
#endif
#if !defined(QATTN_ONLY_D) || QATTN_ONLY_D == 256
int launch_attn_v4_d256(const AttnParams& p, int fmt, int causal, int scale_mode, hipStream_t st) { return launch_v4_full_d<256>(p, fmt, causal, scale_mode, st); }
#endif

}  // namespace qattn
