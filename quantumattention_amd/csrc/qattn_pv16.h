// qattn_pv16.h -- the reference kernel's own P.V numerics for the D = 128 kernel: FP8 QK^T, then 16-bit P and the ORIGINAL 16-bit V.
//
// The reference keeps V and P in 16 bit (src/quantum_attn/tk/attention.py:72 `v_tile` is bf16 / fp16, :286 the exponentiated scores
// are cast to the 16-bit type, :318 the PV product is a 16-bit WGMMA); only Q and K are FP8.  The main path of this build runs both
// GEMMs on FP8 MFMA (north_star), which is accurate where a row's weight is spread over many keys -- and is NOT where a row sees few
// keys: row 0 of a causal head IS V[0], so its error is V's fp8 rounding (0.1 .. 0.25 max-abs at C3 / C5, VERDICT r3 Missing-1).
// pv16_block_pass is the pass for exactly those rows -- query blocks that see fewer than kTwoTermKeys keys (early causal rows, short
// sequences) -- and, as qattn_fp8_attention_forward(v_fmt = QATTN_FMT_BF16 / _FP16), for whole tensors:
//
//   S^T = K.Q^T        v_mfma_f32_32x32x64_f8f6f4, K fragments from the KFRAG image in LDS (as the fp8 passes)
//   P   = exp2(S c - m c)   exact v_exp_f32, running max with a deferred rescale; cast to bf16 / fp16: registers 8s .. 8s+7 of a score
//                      tile ARE the B operand of k-step s; fp32 row sums of the rounded values
//   O^T += V^T.P^T     v_mfma_f32_32x32x16_{bf16,f16}; the A operand comes from the ROW-MAJOR 16-bit V chunk in LDS through
//                      ds_read_b64_tr_b16 (hardware transpose, cdna_hip_programming.md T10): no re-laid copy of V exists anywhere,
//                      the chunk is 64 rows x 256 B = one contiguous 16 KiB of the caller's tensor, copied by LDS-DMA
//
// LDS image of a V chunk: plain 256-byte rows with the 16-byte chunks of row r XOR-ed by f(r) = ((r & 3) << 2) | ((r >> 2) & 3)
// (image (b) of T10): LDS-DMA writes lane-linear, so the swizzle is applied to the SOURCE address (lane i of piece pc fetches chunk
// (i & 15) ^ f(r) of row r = 4 pc + (i >> 4)); with it the transposed reads of a 32-lane half -- 4 rows x 64 bytes -- cover all 64
// banks exactly once.  Keys beyond Skv re-read the last row (their P is 0).
//
// Structure: 8 waves x 32 query rows, a 3-slot ring of {K chunk 8 KiB | V chunk 16 KiB} in the K/V ring's LDS, one barrier per
// 64-key chunk, QK^T -> softmax -> PV per wave in turn (the two waves of a SIMD overlap each other).  This pass serves the short
// early blocks (<= 16 chunks) and the opt-in 16-bit-V mode; it is not software-pipelined like the fp8 sweep.
#pragma once
#include "qattn_attn.h"

namespace qattn {

constexpr int kPv16Slots = 3;
constexpr float kPv16RescaleThr = 5.0f;   // log2 units: P <= 2^5 between rescales (bf16 / fp16 hold it exactly like P <= 1)

typedef short v4s16 __attribute__((ext_vector_type(4)));
typedef __bf16 pv16_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pv16_f16x8 __attribute__((ext_vector_type(8)));

template <int FMT16>
struct Pv16Type;
template <>
struct Pv16Type<QATTN_FMT_BF16> {
    typedef pv16_bf16x8 vec;
    static __device__ __forceinline__ v16f mfma(vec a, vec b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ unsigned pack2(float a, float b) { return pack2_bf16(a, b); }
    static __device__ __forceinline__ float sum2(unsigned w) { return __uint_as_float(w << 16) + __uint_as_float(w & 0xffff0000u); }   // the two ROUNDED values
};
template <>
struct Pv16Type<QATTN_FMT_FP16> {
    typedef pv16_f16x8 vec;
    static __device__ __forceinline__ v16f mfma(vec a, vec b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ unsigned pack2(float a, float b) { return pack2_f16(a, b); }
    static __device__ __forceinline__ float sum2(unsigned w) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 h;
        __builtin_memcpy(&h, &w, 4);
        return (float)h[0] + (float)h[1];
    }
};

// two transposed reads -> the 8 elements of one A operand (keys R .. R+3 and R+8 .. R+11 of this lane's d column)
template <typename Vec>
__device__ __forceinline__ Vec pv16_read_vt(const unsigned char* lo, const unsigned char* hi) {
    const v4s16 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)lo);
    const v4s16 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)hi);
    typedef short v8s16 __attribute__((ext_vector_type(8)));
    const v8s16 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    Vec out;
    __builtin_memcpy(&out, &r, 16);
    return out;
}

// One 256-row query block.  TOKEN: per-row q scales / per-key k scales (standalone entry only); Q16: the fused step's bf16 Q rows,
// quantised here with the pre-pass's quant8 sequence (the same q8 bytes as every other pass of the kernel).
template <int D, int NW, int QK_FMT, int V16_FMT, bool CAUSAL, bool TOKEN, bool Q16, typename DrawIssue, typename DrawFinish>
__device__ __forceinline__ void pv16_block_pass(const AttnParams& p, unsigned char* smem, int tid, int bid, DrawIssue&& draw_issue_hook,
                                                DrawFinish&& draw_finish_hook) {   // hooks around the row stores: the D = 128 kernel requests its next block there
    static_assert((D == 64 || D == 128 || D == 256) && NW == 8, "the DMA split is written for 8 waves");
    static_assert(!Q16 || D == 128, "the fused in-kernel form belongs to the D = 128 kernel");
    typedef Pv16Type<V16_FMT> T;
    typedef typename T::vec vec16;
    constexpr int CH = 64 * D;          // fp8 K chunk
    constexpr int RB = 2 * D;           // bytes of a 16-bit V row
    constexpr int VCH = 64 * RB;        // 16-bit V chunk
    constexpr int STAGE = CH + VCH;
    constexpr int KS = D / 64, MB = D / 32;
    constexpr int KP = CH / 1024, VPW = 2 * KP / NW;   // 1 KiB pieces of a K chunk (4 / 8 / 16); V pieces per wave (1 / 2 / 4)
    constexpr int CPR = RB / 16, RPP = 1024 / RB;      // 16-byte chunks per V row (8 / 16 / 32); rows per piece (8 / 4 / 2)
    static_assert(D != 128 || kPv16Slots * STAGE <= (2 * 2 + 1) * 2 * CH, "D = 128: the ring fits the fp8 sweeps' K/V ring");
    static_assert(kPv16Slots * STAGE <= 160 * 1024, "the ring fits a CU's LDS");
    // The XOR that spreads the transposed reads over the banks, on the 16-byte chunk index of V row r (see the file header for D = 128).
    // A 32-lane half reads 4 rows (r & 3 = 0..3) x 64 bytes; a 64-byte granule covers 16 of the 64 banks, so the four rows' granules must
    // differ mod 4.  Rows are RB bytes apart: D = 128 / 256 (256 / 512 B, = 0 mod 256): XOR the granule index (chunk bits 2..3) with
    // r & 3; D = 64 (128 B: rows r and r + 2 collide): XOR chunk bit 2 with bit 1 of r.
    auto swz = [](int r) -> int { return D == 128 ? (((r & 3) << 2) | ((r >> 2) & 3)) : D == 256 ? ((r & 3) << 2) : (((r >> 1) & 1) << 2); };
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hh = lane >> 5;

    int head, qb;
    map_block(p, bid, p.nqb, CAUSAL, head, qb);
    const int b = head / p.Hq, h = head % p.Hq;
    const long bh = (long)b * p.Hq + h;
    const long kv_head = (long)b * p.Hkv + h / (p.Hq / p.Hkv);
    constexpr int QWG = NW * kQPerWave;
    const int q0_wg = qb * QWG;
    const int q0 = q0_wg + wave * kQPerWave;
    const int qrow = q0 + ql;
    const bool qvalid = qrow < p.Sq;
    const unsigned char* kg = p.k + kv_head * (long)p.nchunks * CH;
    const unsigned char* vg = p.v16 + kv_head * (long)p.Skv * (D * 2);
    const int n_wg = CAUSAL ? min(p.nchunks, (min(q0_wg + QWG, p.Sq) - 1) / 64 + 1) : p.nchunks;
    const int n_w = CAUSAL ? min(n_wg, (q0 + kQPerWave - 1) / 64 + 1) : p.nchunks;

    // ---- one ring stage by LDS-DMA: K chunk t (KP pieces of 1 KiB over the waves) and V rows 64 t .. 64 t + 63 (VPW pieces of RPP rows per wave)
    const int vr = lane / CPR, vc = lane % CPR;                // row within a piece / 16-byte chunk within the row this lane copies
    auto dma_stage = [&](int t, int slot) {
        unsigned char* dst = smem + slot * STAGE;
#pragma unroll
        for (int r = 0; r < (KP + NW - 1) / NW; r++) {
            const int pc = wave + NW * r;                       // (D = 64: waves 0 .. 3 only)
            if (pc < KP) {
                const unsigned char* ksrc = kg + (long)min(t, p.nchunks - 1) * CH + (pc << 10) + (lane << 4);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ksrc,
                                                 (__attribute__((address_space(3))) void*)(dst + (pc << 10)), 16, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < VPW; i++) {
            const int pc = wave * VPW + i;
            const int r = RPP * pc + vr;
            const int key = min(t * 64 + r, p.Skv - 1);
            const int ch = vc ^ swz(r);
            const unsigned char* vsrc = vg + (long)key * RB + (ch << 4);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)vsrc,
                                             (__attribute__((address_space(3))) void*)(dst + CH + (pc << 10)), 16, 0, 0);
        }
    };
    // this wave's pieces per stage: the s_waitcnt immediate that leaves exactly the NEXT stage in flight
    auto wait_stage = [&](bool next_in_flight) {
        if (!next_in_flight) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
        if constexpr (D == 64) {
            if (wave < KP) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        } else if constexpr (D == 128) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    };
    if (n_wg > 0) dma_stage(0, 0);
    if (n_wg > 1) dma_stage(1, 1);

    // ---- Q^T fragments straight into registers: the lane's 32 bytes d = 64 s + 32 hh .. + 31 of its row
    v8i qf[KS];
    float c;
    if (Q16) {
        const float inv_qmax = (float)(1.0 / (double)(QK_FMT == QATTN_FMT_E4M3 ? 448.0 : 57344.0));
        const float scale_q = make_scale(__uint_as_float(max_partials(p.q_amax_part + bh * p.amax_stride, p.amax_n, lane) & 0x7fffffffu), inv_qmax,
                                         p.q_numerics, QATTN_FMT_BF16);
        if (q0_wg == 0 && tid == 0) p.sq_out[bh] = scale_q;   // (the block that holds row 0 writes the head's scale, whichever pass runs it)
        const float rinv = 1.0f / scale_q;
        const uint4* qp = reinterpret_cast<const uint4*>(p.q16 + ((bh * p.Sq + (qvalid ? qrow : 0)) * D + hh * 32) * 2);
#pragma unroll
        for (int s = 0; s < KS; s++) {
            int2 w[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                uint4 raw = qp[s * 8 + i];
                if (!qvalid) raw = make_uint4(0, 0, 0, 0);
                w[i] = quant8<QATTN_FMT_BF16, QK_FMT>(raw, scale_q, rinv);
            }
            qf[s] = v8i{w[0].x, w[0].y, w[1].x, w[1].y, w[2].x, w[2].y, w[3].x, w[3].y};
        }
        c = p.sm_log2e * scale_q * p.sk[kv_head];
    } else {
        const unsigned char* qp = p.q + ((bh * p.Sq + (qvalid ? qrow : 0)) * D) + hh * 32;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            v4i lo = *reinterpret_cast<const v4i*>(qp + s * 64), hi = *reinterpret_cast<const v4i*>(qp + s * 64 + 16);
            if (!qvalid) { lo = v4i{0, 0, 0, 0}; hi = v4i{0, 0, 0, 0}; }
            qf[s] = v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
        c = TOKEN ? p.sm_log2e * (qvalid ? p.sq[bh * p.Sq + qrow] : 1.0f) : p.sm_log2e * p.sq[bh] * p.sk[kv_head];
    }
    const float* skt = TOKEN ? p.sk + kv_head * p.Skv : nullptr;
    // ---- per-lane pieces of the transposed-read addresses (T10): lane 4 q4 + p4 of a 16-lane group supplies row R + q4, the 8 bytes
    // at element 4 p4 of the group's 16 columns; the group's columns are d = 32 m + 16 cg .. + 15 (cg = group & 1), its rows start at
    // R = 32 tt + 16 s + 4 hh (elements 0..3 of the operand) and R + 8 (elements 4..7) -- the key order in which the S^T accumulator
    // registers 8 s .. 8 s + 7 become a B operand.  R & 3 = 0 and (R >> 2) & 3 = hh resp. hh + 2, so f(row) = (q4 << 2) | hh [+ 2].
    const int q4 = (lane >> 2) & 3, p4 = lane & 3, cg = (lane >> 4) & 1;
    const int cc = 2 * cg + (p4 >> 1);
    // (D = 128: the low chunk bits take (R >> 2) & 3 = hh resp. hh + 2; the other images leave them alone)
    const unsigned tr_lo = (unsigned)RB * (4 * hh + q4) + 16u * (cc ^ (D == 128 ? hh : 0)) + 8u * (p4 & 1);
    const unsigned tr_hi = (unsigned)RB * (4 * hh + 8 + q4) + 16u * (cc ^ (D == 128 ? hh + 2 : 0)) + 8u * (p4 & 1);
    const int qsw = D == 64 ? (q4 >> 1) : q4;                   // what the granule index m is XOR-ed with (swz above, on rows R + q4)
    const int frag_lane_off = (hh << 10) + (ql << 4);

    v16f o[MB];
#pragma unroll
    for (int m = 0; m < MB; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[m][r] = 0.0f;
    float m_run = -1.0e30f, l_run = 0.0f;

    int slot = 0;
    for (int t = 0; t < n_wg; t++) {
        // stage t has landed (this wave's three pieces; the stage behind it may still be in flight), then everyone's pieces are visible
        // and every wave has left the slot that stage t + 2 is about to overwrite
        wait_stage(t + 1 < n_wg);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 2 < n_wg) dma_stage(t + 2, slot >= 1 ? slot - 1 : slot + 2);
        if (t < n_w) {   // wave-uniform (causal: waves whose rows end earlier keep the barrier cadence)
            const unsigned char* kbuf = smem + slot * STAGE + frag_lane_off;
            const unsigned char* vbuf = smem + slot * STAGE + CH;
            v16f s0, s1;
#pragma unroll
            for (int r = 0; r < 16; r++) { s0[r] = 0.0f; s1[r] = 0.0f; }
#pragma unroll
            for (int s = 0; s < KS; s++) {
                const v8i ka = lds_read_frag(kbuf + ((0 * KS + s) << 11)), kb = lds_read_frag(kbuf + ((1 * KS + s) << 11));
                s0 = mfma_f8<QK_FMT, QK_FMT>(ka, qf[s], s0);
                s1 = mfma_f8<QK_FMT, QK_FMT>(kb, qf[s], s1);
            }
            if constexpr (TOKEN) {
                // per-key scales: registers 4 j .. 4 j + 3 of tile tt hold keys t 64 + 32 tt + 8 j + 4 hh .. + 3.  The 8 scales of (tt, j) sit at
                // a wave-uniform address -- scalar loads, the lane's half picked by hh -- so they cost no vector registers (at D = 256 the
                // pass has none to spare: 128 of O^T, 32 of Q^T, 32 of scores, 16 of P).  Keys beyond Skv: the last scale (masked below).
#pragma unroll
                for (int tt = 0; tt < 2; tt++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int kb = t * 64 + 32 * tt + 8 * j;
                        const int last = p.Skv - 1;
                        v16f& sx = tt ? s1 : s0;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const float wa = skt[min(kb + i, last)], wb = skt[min(kb + 4 + i, last)];
                            sx[4 * j + i] *= hh ? wb : wa;
                        }
                    }
            }
            prep_scores<CAUSAL, false>(s0, s1, p, t * 64, q0, qrow, hh, nullptr);
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
            if (__any((mx - m_run) * c > kPv16RescaleThr)) {
                const float m_new = fmaxf(m_run, mx);
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
#pragma unroll
                for (int m = 0; m < MB; m++)
#pragma unroll
                    for (int r = 0; r < 16; r++) o[m][r] *= alpha;
                l_run *= alpha;
                m_run = m_new;
            }
            const float mc = -m_run * c;
            // P: fp32 exponentials cast pairwise; pb[tt][s] = the B operand of k-step s of tile tt.  The row sum adds the ROUNDED values,
            // so that numerator and denominator see the same weights and a row carried by one key reproduces that key's V row to the
            // output rounding whatever the deferred reference is (the reference sums the un-rounded values, tk/attention.py:297-301;
            // with its exact running max the top key's P is exactly 1 and the two agree there)
            vec16 pb[2][2];
            float ls = 0.0f;
#pragma unroll
            for (int tt = 0; tt < 2; tt++)
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    const v16f& sx = tt ? s1 : s0;
                    unsigned w[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[8 * s + 2 * j], c, mc));
                        const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[8 * s + 2 * j + 1], c, mc));
                        w[j] = T::pack2(e0, e1);
                        ls += T::sum2(w[j]);
                    }
                    const v4i wv = {(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
                    __builtin_memcpy(&pb[tt][s], &wv, 16);
                }
            l_run += ls;
#pragma unroll
            for (int m = 0; m < MB; m++) {
                const unsigned xm = 64u * (unsigned)(m ^ qsw);
                const unsigned char* alo = vbuf + tr_lo + xm;
                const unsigned char* ahi = vbuf + tr_hi + xm;
#pragma unroll
                for (int tt = 0; tt < 2; tt++)
#pragma unroll
                    for (int s = 0; s < 2; s++) {
                        const int roff = RB * (32 * tt + 16 * s);
                        o[m] = T::mfma(pv16_read_vt<vec16>(alo + roff, ahi + roff), pb[tt][s], o[m]);
                    }
            }
        }
        slot = slot == kPv16Slots - 1 ? 0 : slot + 1;
    }
    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    const float l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    const unsigned ticket = draw_issue_hook();
    store_o_rows<MB>(p.out, p.out_fmt, o, 1.0f / l_tot, bh * p.Sq + qrow, hh, qvalid);
    draw_finish_hook(ticket);
    if (p.lse && hh == 0 && qvalid) p.lse[bh * p.lse_stride + qrow] = (0.6931471805599453f * (m_run * c) + __logf(l_tot)) * p.lse_mul;
}

// The 16-bit-V form of qattn_fp8_attention_forward (v_fmt = QATTN_FMT_BF16 / _FP16): every query block through pv16_block_pass, one
// workgroup per block (map_block: XCD-contiguous heads, causal blocks heaviest first).
// n_blocks > 0: only the first n_blocks query blocks of every head (the fused step's early rows on the paths whose main kernel has no
// 16-bit-V pass of its own: token-wise scales, fp16 inputs).
int launch_attn_pv16(const AttnParams& p, int D, int qk_fmt, int v16_fmt, int causal, int scale_mode, hipStream_t st, int n_blocks = 0);
// leading query blocks (of 256 rows) whose first row sees fewer than two_term_keys keys
inline int pv16_early_blocks(int Sq, int Skv, int causal, int two_term_keys) {
    // block qb sees Skv keys (non-causal) or min(Skv, 256 qb + 1): early while that is below the threshold
    const int nqb = ceil_div(Sq, kQPerWG);
    if (Skv < two_term_keys) return nqb;
    if (!causal) return 0;
    const int cnt = (two_term_keys - 1 + kQPerWG - 1) / kQPerWG;
    return cnt < nqb ? cnt : nqb;
}

}  // namespace qattn
