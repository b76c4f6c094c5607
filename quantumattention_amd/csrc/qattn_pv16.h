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
    static __device__ __forceinline__ v4f mfma_sum(vec a, vec b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static constexpr unsigned kOnes = 0x3f803f80u;   // two 1.0
};
template <>
struct Pv16Type<QATTN_FMT_FP16> {
    typedef pv16_f16x8 vec;
    static __device__ __forceinline__ v16f mfma(vec a, vec b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ unsigned pack2(float a, float b) { return pack2_f16(a, b); }
    static __device__ __forceinline__ v4f mfma_sum(vec a, vec b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static constexpr unsigned kOnes = 0x3c003c00u;
};

// two transposed reads -> the 8 elements of one A operand (keys R .. R+3 and R+8 .. R+11 of this lane's d column)
// Issued through asm: behind the builtin (a known LDS load) the compiler waits for EVERY LDS-DMA in flight first -- it cannot tell that the
// ring slot being read is not the one being filled -- which drained the stage requested a moment earlier at every chunk (s_waitcnt vmcnt(0)
// in front of the first transposed read).  The ring protocol's own counted wait and barrier are what order these reads; the caller waits
// for the data with pv16_wait_lds<N> (N = transposed reads issued after the ones it needs).
typedef int v2i32 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2i32 pv16_read_tr(unsigned addr) {
    v2i32 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
template <int OFF>
__device__ __forceinline__ v2i32 pv16_read_tr_at(unsigned addr) {
    v2i32 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
// the eight reads of one 32-column block of V^T (operands of its four products) are awaited together; the registers are operands of
// the wait so that no product is scheduled above it
template <int N>
__device__ __forceinline__ void pv16_wait_lds(v2i32 (&r)[8]) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                 : "n"(N)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void pv16_wait_lds(v2i32 (&r)[8], v2i32 (&q)[8]) {
    asm volatile("s_waitcnt lgkmcnt(%16)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(q[0]), "+v"(q[1]), "+v"(q[2]),
                   "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])
                 : "n"(N)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void pv16_wait_lds(v2i32 (&r)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(N) : "memory");
}
template <typename Vec>
__device__ __forceinline__ Vec pv16_operand(v2i32 lo, v2i32 hi) {
    const v4i r = {lo[0], lo[1], hi[0], hi[1]};
    Vec out;
    __builtin_memcpy(&out, &r, 16);
    return out;
}

// One 256-row query block.  TOKEN: per-row q scales / per-key k scales (standalone entry only); Q16: the fused step's bf16 Q rows,
// quantised here with the pre-pass's quant8 sequence (the same q8 bytes as every other pass of the kernel).
template <int D, int NW, int QK_FMT, int V16_FMT, bool CAUSAL, bool TOKEN, bool Q16, int NS = kPv16Slots, bool PP = false, typename DrawIssue, typename DrawFinish>
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
    static_assert(NS >= 3 && NS <= 4, "ring depth: see the two loop forms below");
    static_assert(NS * STAGE <= 160 * 1024, "the ring fits a CU's LDS");
    // PP: the two-group loop (below) instead of the one-group loop.  It pays on long sweeps at D = 128 (whole-tensor launches: C2 shape 0.823
    // against 0.836 ms, S = 16384 2.97 against 3.07) and costs on the short ones of the fused step's early rows (two idle half-steps per
    // block: C3 step +1 %), which therefore keep the one-group loop; D = 64 runs two workgroups per CU in the one-group loop (125
    // registers), which the two-group loop's 139 do not allow (0.45 against 0.53 ms); D = 256 has no registers for it.
    // profiles/r04/time_16bit_v_mode_loop_forms.log, ab_c3_pv16_loop_forms.log
    static_assert(!PP || D == 128, "the two-group loop is instantiated for D = 128");
    constexpr int PW = (KP + NW - 1) / NW + VPW;        // LDS-DMA pieces per wave and stage: 2 / 3 / 6 (D = 64: see dma_stage)
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
    const unsigned char* vg = v16_head(p, b, h / (p.Hq / p.Hkv), kv_head, RB);   // (the caller's V may be a strided view: rows vrs bytes apart)
    const long vrs = v16_row_stride(p, RB);
    const int n_wg = CAUSAL ? min(p.nchunks, (min(q0_wg + QWG, p.Sq) - 1) / 64 + 1) : p.nchunks;
    const int n_w = CAUSAL ? min(n_wg, (q0 + kQPerWave - 1) / 64 + 1) : p.nchunks;

    // ---- one ring stage by LDS-DMA: K chunk t (KP pieces of 1 KiB over the waves) and V rows 64 t .. 64 t + 63 (VPW pieces of RPP rows per wave)
    const int vr = lane / CPR, vc = lane % CPR;                // row within a piece / 16-byte chunk within the row this lane copies
    // (per-lane source addresses of the wave's pieces within chunk 0, computed once per block: per stage a uniform offset is added)
    constexpr int KPW = (KP + NW - 1) / NW;
    const unsigned char* ksrc0[KPW];
    const unsigned char* vsrc0[VPW];
#pragma unroll
    for (int r = 0; r < KPW; r++) {
        // (D = 64: four pieces for eight waves -- waves 4 .. 7 fetch them once more, the same bytes to the same place: every wave then has
        // the same number of requests in flight, which is what the counted waits below assume)
        const int pc = (wave + NW * r) % KP;
        ksrc0[r] = kg + (pc << 10) + (lane << 4);
    }
#pragma unroll
    for (int i = 0; i < VPW; i++) {
        const int r = RPP * (wave * VPW + i) + vr;
        vsrc0[i] = vg + (long)r * vrs + ((vc ^ swz(r)) << 4);
    }
    // piece i of the wave's PW pieces of stage t (K pieces first), into ring slot `slot`
    auto dma_piece = [&](int i, int t, int slot) {
        unsigned char* dst = smem + slot * STAGE;
        if (i < KPW) {
            const int pc = (wave + NW * i) % KP;
            const unsigned char* ksrc = ksrc0[i < KPW ? i : 0] + (long)min(t, p.nchunks - 1) * CH;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ksrc,
                                             (__attribute__((address_space(3))) void*)(dst + (pc << 10)), 16, 0, 0);
        } else {
            const int j = i - KPW, pc = wave * VPW + j;
            const unsigned char* vsrc = vsrc0[j >= 0 && j < VPW ? j : 0] + (long)t * (64 * vrs);
            if (t * 64 + 64 > p.Skv) {   // (workgroup-uniform) the head's last, ragged chunk: keys beyond Skv re-read the last row
                const int r = RPP * pc + vr;
                vsrc = vg + (long)min(t * 64 + r, p.Skv - 1) * vrs + ((vc ^ swz(r)) << 4);
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)vsrc,
                                             (__attribute__((address_space(3))) void*)(dst + CH + (pc << 10)), 16, 0, 0);
        }
    };
    auto dma_stage = [&](int t, int slot) {
#pragma unroll
        for (int i = 0; i < PW; i++) dma_piece(i, t, slot);
    };
    // waits until at most k STAGES requested after the one needed are still in flight (k workgroup-uniform; PW requests per wave and stage)
    auto wait_younger = [&](int k) {
        if (k <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (k == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
        else if (k == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PW) : "memory");
    };
    // ---- Q^T fragments straight into registers: the lane's 32 bytes d = 64 s + 32 hh .. + 31 of its row
    v8i qf[KS];
    float c;
    if (Q16) {
        const float inv_qmax = (float)(1.0 / (double)(QK_FMT == QATTN_FMT_E4M3 ? 448.0 : 57344.0));
        const float scale_q = make_scale(__uint_as_float(max_partials(p.q_amax_part + bh * p.amax_stride, p.amax_n, lane) & 0x7fffffffu), inv_qmax,
                                         p.q_numerics, V16_FMT);   // (the fused step's q, k, v share one 16-bit type)
        if (q0_wg == 0 && tid == 0) p.sq_out[bh] = scale_q;   // (the block that holds row 0 writes the head's scale, whichever pass runs it)
        const float rinv = 1.0f / scale_q;
        const uint4* qp = reinterpret_cast<const uint4*>(q16_row(p, b, h, bh, qvalid ? qrow : 0, D * 2) + hh * 64);
#pragma unroll
        for (int s = 0; s < KS; s++) {
            int2 w[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                uint4 raw = qp[s * 8 + i];
                if (!qvalid) raw = make_uint4(0, 0, 0, 0);
                w[i] = quant8<V16_FMT, QK_FMT>(raw, scale_q, rinv);
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
    // the ring's first stages, requested behind the Q loads (the counter is in order: waiting for a stage never waits for Q's successors)
    int issued = 0;
    for (; issued < (PP ? NS : 2) && issued < n_wg; issued++) dma_stage(issued, issued);
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
    float m_run = -1.0e30f;
    // Row sums of the ROUNDED P on the matrix pipe: ones(16 x 32) . P^T with one v_mfma_f32_16x16x32 per k-step.  Read as that shape's B
    // operand, a k-step's P registers are 16 queries x four 8-key groups {lanes 0-15: queries 0-15, keys 0-7 | 16-31: queries 16-31, keys
    // 0-7 | 32-47: queries 0-15, keys 8-15 | 48-63: queries 16-31, keys 8-15}; A row 0 is one on the even groups, row 1 on the odd ones, so
    // output row 0 (register 0 of lanes 0-15) sums query n's 16 keys and row 1 (register 1) query n + 16's (WaveState::lsum of the fp8
    // sweeps, qattn_attn_v2.hip).  32 unpack-and-add pairs per chunk and lane become four 16-cycle products.
    v4f lsum = {0.0f, 0.0f, 0.0f, 0.0f};
    vec16 ones;
    {
        const int row = lane & 15, kgrp = lane >> 4;
        const int one = ((row == 0 && !(kgrp & 1)) || (row == 1 && (kgrp & 1))) ? (int)T::kOnes : 0;
        const v4i w = {one, one, one, one};
        __builtin_memcpy(&ones, &w, 16);
    }

    // ---- the three parts of a chunk's work
    v16f s0, s1;        // S^T of the chunk between its QK^T and its softmax
    vec16 pb[2][2];     // P of the chunk between its softmax and its PV: pb[tt][s] = the B operand of k-step s of tile tt
    auto qk = [&](int t, int slot) {   // S^T(t) = K(t).Q^T
        const unsigned char* kbuf = smem + slot * STAGE + frag_lane_off;
#pragma unroll
        for (int r = 0; r < 16; r++) { s0[r] = 0.0f; s1[r] = 0.0f; }
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const v8i ka = lds_read_frag(kbuf + ((0 * KS + s) << 11)), kb = lds_read_frag(kbuf + ((1 * KS + s) << 11));
            s0 = mfma_f8<QK_FMT, QK_FMT>(ka, qf[s], s0);
            s1 = mfma_f8<QK_FMT, QK_FMT>(kb, qf[s], s1);
        }
    };
    auto no_hook = [](int) {};
    auto softmax = [&](int t, auto&& hook) {        // P(t) from S^T(t); may rescale O^T and the row sums.  hook(0 .. 2): three points spread over it
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
            lsum[0] *= alpha;                                                        // lane n < 16: query n ...
            lsum[1] *= __uint_as_float(swizzle_xor16(__float_as_uint(alpha)));       // ... and query n + 16 (lane n + 16's factor)
            m_run = m_new;
        }
        const float mc = -m_run * c;
        hook(0);
        // P: fp32 exponentials cast pairwise; pb[tt][s] = the B operand of k-step s of tile tt.  The row sum adds the ROUNDED values,
        // so that numerator and denominator see the same weights and a row carried by one key reproduces that key's V row to the
        // output rounding whatever the deferred reference is (the reference sums the un-rounded values, tk/attention.py:297-301;
        // with its exact running max the top key's P is exactly 1 and the two agree there)
#pragma unroll
        for (int tt = 0; tt < 2; tt++)
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const v16f& sx = tt ? s1 : s0;
                unsigned w[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    // (two v_fma_f32, not one v_pk_fma_f32: the packed form costs several issue slots -- MI355X_MICROARCH.md, constants table)
                    const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[8 * s + 2 * j], c, mc));
                    const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[8 * s + 2 * j + 1], c, mc));
                    w[j] = T::pack2(e0, e1);
                }
                const v4i wv = {(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
                __builtin_memcpy(&pb[tt][s], &wv, 16);
                if (s == 1) hook(1 + tt);
            }
    };
    auto pv = [&](int slot) {          // O^T += V^T.P^T and the row sums, V from ring slot `slot`
#pragma unroll
        for (int tt = 0; tt < 2; tt++)
#pragma unroll
            for (int s = 0; s < 2; s++) lsum = T::mfma_sum(ones, pb[tt][s], lsum);
        const unsigned vaddr = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)(smem + slot * STAGE + CH);
        if constexpr (D <= 128) {
            // O^T += V^T.P^T, one k-step (16 keys) after the other over all MB column blocks: consecutive products go to DIFFERENT
            // accumulators (four products in a row into one accumulator each waited for its predecessor's result: the products of a
            // chunk took twice their pipe time), and the next k-step's transposed reads travel meanwhile
            unsigned alo[MB], ahi[MB];
#pragma unroll
            for (int m = 0; m < MB; m++) {
                const unsigned xm = 64u * (unsigned)(m ^ qsw);
                alo[m] = vaddr + tr_lo + xm;
                ahi[m] = vaddr + tr_hi + xm;
            }
            auto issue = [&](v2i32 (&r)[2 * MB], auto j_tag) {   // k-step j = 2 tt + s: keys 16 j .. 16 j + 15 of the chunk
                constexpr int J = decltype(j_tag)::value;
#pragma unroll
                for (int m = 0; m < MB; m++) {
                    r[2 * m] = pv16_read_tr_at<RB * 16 * J>(alo[m]);
                    r[2 * m + 1] = pv16_read_tr_at<RB * 16 * J>(ahi[m]);
                }
            };
            auto multiply = [&](v2i32 (&r)[2 * MB], const vec16& pj) {
#pragma unroll
                for (int m = 0; m < MB; m++) o[m] = T::mfma(pv16_operand<vec16>(r[2 * m], r[2 * m + 1]), pj, o[m]);
            };
            using J0 = std::integral_constant<int, 0>; using J1 = std::integral_constant<int, 1>;
            using J2 = std::integral_constant<int, 2>; using J3 = std::integral_constant<int, 3>;
            v2i32 ra[2 * MB], rb[2 * MB];
            issue(ra, J0{});
            issue(rb, J1{});
            pv16_wait_lds<2 * MB>(ra);
            multiply(ra, pb[0][0]);
            issue(ra, J2{});
            pv16_wait_lds<2 * MB>(rb);
            multiply(rb, pb[0][1]);
            issue(rb, J3{});
            pv16_wait_lds<2 * MB>(ra);
            multiply(ra, pb[1][0]);
            pv16_wait_lds<0>(rb);
            multiply(rb, pb[1][1]);
        } else {
            // D = 256: no registers for a second set of operands: one 32-column block of V^T after the other -- read, wait, multiply
            v2i32 ra[8];
#pragma unroll
            for (int m = 0; m < MB; m++) {
                const unsigned xm = 64u * (unsigned)(m ^ qsw);
                const unsigned alo = vaddr + tr_lo + xm, ahi = vaddr + tr_hi + xm;
                ra[0] = pv16_read_tr_at<RB * 0>(alo);  ra[1] = pv16_read_tr_at<RB * 0>(ahi);
                ra[2] = pv16_read_tr_at<RB * 16>(alo); ra[3] = pv16_read_tr_at<RB * 16>(ahi);
                ra[4] = pv16_read_tr_at<RB * 32>(alo); ra[5] = pv16_read_tr_at<RB * 32>(ahi);
                ra[6] = pv16_read_tr_at<RB * 48>(alo); ra[7] = pv16_read_tr_at<RB * 48>(ahi);
                pv16_wait_lds<0>(ra);
                o[m] = T::mfma(pv16_operand<vec16>(ra[0], ra[1]), pb[0][0], o[m]);
                o[m] = T::mfma(pv16_operand<vec16>(ra[2], ra[3]), pb[0][1], o[m]);
                o[m] = T::mfma(pv16_operand<vec16>(ra[4], ra[5]), pb[1][0], o[m]);
                o[m] = T::mfma(pv16_operand<vec16>(ra[6], ra[7]), pb[1][1], o[m]);
            }
        }
    };

    if constexpr (!PP) {
        // ---- one group: every wave runs QK^T -> softmax -> PV on chunk t between two barriers (3-slot ring, two stages ahead)
        int slot = 0;
        for (int t = 0; t < n_wg; t++) {
            // stage t has landed (the stage behind it may still be in flight), then everyone's pieces are visible and every wave has left
            // the slot that stage t + 2 is about to overwrite
            wait_younger(issued - 1 - t);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (issued < n_wg) { dma_stage(issued, issued % NS); issued++; }
            if (t < n_w) {   // wave-uniform (causal: waves whose rows end earlier keep the barrier cadence)
                qk(t, slot);
                softmax(t, no_hook);
                pv(slot);
            }
            slot = slot == NS - 1 ? 0 : slot + 1;
        }
    } else {
        // ---- two groups of four waves, half a chunk apart (the two waves of a SIMD are one of each: waves w and w + 4).  Between two
        // barriers one group runs its matrix products -- PV(t - 1) and QK^T(t) -- while the other runs its softmax: with every wave in the
        // same phase, as in the one-group form, both waves of a SIMD want the matrix pipe, then both want the vector pipe, and the
        // passes of a block ran at the SUM of the two (D = 128, C2 shape: 2930 cycles per chunk for 1536 of products).
        //   half-step 2u    : group 0: PV(u-1), QK(u)      group 1: softmax(u-1)
        //   half-step 2u + 1: group 0: softmax(u)          group 1: PV(u-1), QK(u)
        // Stage t = {K(t), V(t)} is read from half-step 2t (group 0's QK) to 2t + 3 (group 1's PV): its slot takes stage t + NS at
        // half-step 2t + 4, which must have landed by 2t + 2 NS.
        // (waves w and w + 4 share a SIMD: with the groups cut as wave & 1 or (wave >> 1) & 1 the same pass took 1.15 / 1.04 ms against 0.91)
        const int grp = wave >> 2;
        // PV(u - 1) and QK^T(u) as one sequence: all a trip's LDS latency in one place.  The first two k-steps' transposed reads and the K
        // fragments are requested together; QK^T runs as soon as K is there (the row-sum products cover part of that wait), the other two
        // k-steps' reads are requested behind it and land under the first eight PV products.  (Requested one k-step ahead, as pv() does it,
        // every k-step waited ~150 cycles for its operands: 1400 cycles per trip for 830 of products -- dev stamps, profiles/r04/pv16_phase_stamps.log.)
        auto pv_qk = [&](int slot_v, int slot_k) {
            static_assert(MB == 4, "register sets of eight transposed reads");
            const unsigned vaddr = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)(smem + slot_v * STAGE + CH);
            unsigned alo[MB], ahi[MB];
#pragma unroll
            for (int m = 0; m < MB; m++) {
                const unsigned xm = 64u * (unsigned)(m ^ qsw);
                alo[m] = vaddr + tr_lo + xm;
                ahi[m] = vaddr + tr_hi + xm;
            }
            v2i32 ra[8], rb[8], rc[8];   // (a fourth set does not fit 256 registers: the last k-step re-uses the first one's, once its products are long gone)
#pragma unroll
            for (int m = 0; m < MB; m++) { ra[2 * m] = pv16_read_tr_at<RB * 0>(alo[m]); ra[2 * m + 1] = pv16_read_tr_at<RB * 0>(ahi[m]); }
#pragma unroll
            for (int m = 0; m < MB; m++) { rb[2 * m] = pv16_read_tr_at<RB * 16>(alo[m]); rb[2 * m + 1] = pv16_read_tr_at<RB * 16>(ahi[m]); }
#pragma unroll
            for (int tt = 0; tt < 2; tt++)
#pragma unroll
                for (int s = 0; s < 2; s++) lsum = T::mfma_sum(ones, pb[tt][s], lsum);
            qk(0, slot_k);   // (the chunk index is not used by qk)
#pragma unroll
            for (int m = 0; m < MB; m++) { rc[2 * m] = pv16_read_tr_at<RB * 32>(alo[m]); rc[2 * m + 1] = pv16_read_tr_at<RB * 32>(ahi[m]); }
            pv16_wait_lds<8>(ra, rb);
#pragma unroll
            for (int m = 0; m < MB; m++) o[m] = T::mfma(pv16_operand<vec16>(ra[2 * m], ra[2 * m + 1]), pb[0][0], o[m]);
#pragma unroll
            for (int m = 0; m < MB; m++) o[m] = T::mfma(pv16_operand<vec16>(rb[2 * m], rb[2 * m + 1]), pb[0][1], o[m]);
            // (k-step 0's four products were issued 128 pipe cycles ago: their operands have been read)
#pragma unroll
            for (int m = 0; m < MB; m++) { ra[2 * m] = pv16_read_tr_at<RB * 48>(alo[m]); ra[2 * m + 1] = pv16_read_tr_at<RB * 48>(ahi[m]); }
            pv16_wait_lds<8>(rc);
#pragma unroll
            for (int m = 0; m < MB; m++) o[m] = T::mfma(pv16_operand<vec16>(rc[2 * m], rc[2 * m + 1]), pb[1][0], o[m]);
            pv16_wait_lds<0>(ra);
#pragma unroll
            for (int m = 0; m < MB; m++) o[m] = T::mfma(pv16_operand<vec16>(ra[2 * m], ra[2 * m + 1]), pb[1][1], o[m]);
        };
        // One code path for every trip: the first one multiplies chunk 0's V by P = 0 (pb starts as zeros), the wave's last one computes a
        // QK^T nobody reads (on whatever the slot holds).  Separate PV-only / QK^T-only paths cost registers the loop does not have.
        auto products = [&](int u) {
            if (u <= n_w) pv_qk(u >= 1 ? (u - 1) % NS : 0, u % NS);
        };
        {
            const v4i z = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 4; i++) __builtin_memcpy(&pb[i >> 1][i & 1], &z, 16);
        }
        wait_younger(issued - 1);
        __builtin_amdgcn_s_barrier();
        // (two loops, one per group, each a straight sequence of its two phases: with one loop and the phase picked inside it the
        // accumulators met at the join of the two branches and the compiler copied them around -- 400 register moves per trip, spills)
        // Each group requests its pieces of the next stage at the head of its SOFTMAX phase (the matrix pipe does not wait for the 360 .. 660
        // issue cycles of three requests, dev stamps).  Spread over the phase -- one request after each third of the exponentials -- they
        // cost MORE: the phase grew from 1520 to 1980 cycles; behind the last product of the products phase (the shorter one) they cost 500 ..
        // 670 and the pass got 3 % slower (profiles/r04/pv16_phase_stamps.log).
        auto softmax_and_dma = [&](int u, bool do_softmax, int t) {
            asm volatile("" ::: "memory");
            if (u >= 2 && issued < n_wg) { dma_stage(issued, issued % NS); issued++; }   // (stage u - 2's slot: free since the last barrier but one)
            if (do_softmax) softmax(t, no_hook);
        };
        auto mid = [&]() {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        auto tail = [&](int u) {
            if (u + 1 < n_wg) wait_younger(issued - 2 - u);   // stage u + 1, needed from the next half-step on
            __builtin_amdgcn_s_barrier();
        };
#ifdef QATTN_PV16_STAMP   // development: cycles per phase, summed over the sweep, written over the wave's first LSE entries
        unsigned long long acc[5] = {0, 0, 0, 0, 0}, tl = __builtin_amdgcn_s_memtime();
#define PV16_T(I) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc[I] += t_ - tl; tl = t_; } while (0)
#else
#define PV16_T(I) do { } while (0)
#endif
        if (grp == 0) {
#pragma nounroll
            for (int u = 0; u <= n_wg; u++) {
                products(u);
                PV16_T(0);
                mid();
                PV16_T(2);
                softmax_and_dma(u, u < n_w, u);
                PV16_T(1);
                tail(u);
                PV16_T(3);
            }
        } else {
#pragma nounroll
            for (int u = 0; u <= n_wg; u++) {
                softmax_and_dma(u, u >= 1 && u - 1 < n_w, u - 1);
                PV16_T(1);
                mid();
                PV16_T(2);
                products(u);
                PV16_T(0);
                tail(u);
                PV16_T(3);
            }
        }
#ifdef QATTN_PV16_STAMP
        if (p.lse && lane == 0 && q0 + 8 <= p.Sq)
            for (int i = 0; i < 5; i++) p.lse[bh * p.lse_stride + q0 + 1 + i] = (float)acc[i];
#endif
#undef PV16_T
    }
    // query q's sum sits in lane q & 15, register q >> 4 (both half-waves' keys already added by the MFMA)
    const float l_lo = bcast_low16(lsum[0]), l_hi = bcast_low16(lsum[1]);
    const float l_tot = (lane & 16) ? l_hi : l_lo;
    const unsigned ticket = draw_issue_hook();
    store_o_rows<MB>(p.out, p.out_fmt, o, 1.0f / l_tot, out_row_offset(p, bh, qrow, MB * 64), hh, qvalid);
    draw_finish_hook(ticket);
#ifndef QATTN_PV16_STAMP
    if (p.lse && hh == 0 && qvalid) p.lse[bh * p.lse_stride + qrow] = (0.6931471805599453f * (m_run * c) + __logf(l_tot)) * p.lse_mul;
#endif
    if (p.path && hh == 0 && qvalid) p.path[bh * p.Sq + qrow] = (unsigned char)QATTN_PATH_V16;
}

// ---------------------------------------------------------------------------------------------------------
// The row-level rescue on the reference's own P.V numerics (round 5): the 32 gathered rows of rescue_rows_at
// (qattn_attn.h) recomputed with FP8 QK^T, exact exponentials, 16-bit P and the ORIGINAL 16-bit V, the key range split over the NW = 8
// waves of the block as there.  A flagged row is one whose weight sits on few keys: exactly the rows whose output carries V's rounding
// error one to one -- two-term P cured P's rounding and left V's 2^-4 (0.045 max-abs against fp64 SDPA on the 16-bit V at a score
// spread of 2, VERDICT r4 Missing-2).
// Every wave copies ITS chunk of V (64 rows x 256 B, the XOR image of pv16_block_pass) into a 16 KiB area of its own by LDS-DMA
// and reads the K fragments straight from global memory / L2 (both requests of a chunk travel together); one chunk in flight per wave:
// a rescue is a handful of chunks per wave and latency-bound either way.  LDS: 8 x 16 KiB of V, the gathered rows' parked Q^T
// fragments behind them (the caller's), the merge slots alias the V areas afterwards.
// ---------------------------------------------------------------------------------------------------------
constexpr int kRescue16VBytes = 8 * 64 * 2 * 128;   // the V areas of the eight waves (D = 128)
template <int D, int NW, int QK_FMT, int V16_FMT, bool CAUSAL, typename QFrag>
__device__ __forceinline__ void rescue_rows16_at(const AttnParams& p, unsigned char* smem, const unsigned char* kg, const unsigned char* vg16,
                                                 int row, bool store, int row_lo, int row_hi, int wave, int lane, long bh, float c, QFrag&& qfrag) {
    static_assert(NW == 8 && D == 128, "eight 16 KiB V areas, three merge rounds");
    typedef Pv16Type<V16_FMT> T;
    typedef typename T::vec vec16;
    constexpr int CH = 64 * D, KS = D / 64, MB = D / 32, RB = 2 * D, VCH = 64 * RB;
    constexpr int SLOT = rescue_slot_bytes<D>();
    static_assert(NW * VCH == kRescue16VBytes && 4 * SLOT <= NW * VCH, "the merge slots alias the V areas");
    const int ql = lane & 31, hh = lane >> 5;
    const int frag_lane_off = (hh << 10) + (ql << 4);
    v8i qf[KS];
#pragma unroll
    for (int s = 0; s < KS; s++) qf[s] = qfrag(s);
    const int n_r = CAUSAL ? min(p.nchunks, min(row_hi, p.Sq - 1) / 64 + 1) : p.nchunks;
    const int per = (n_r + NW - 1) / NW;
    const int t0 = wave * per, t1 = min(n_r, t0 + per);
    v16f o[MB];
#pragma unroll
    for (int m = 0; m < MB; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[m][r] = 0.0f;
    float m_run = -1.0e30f;
    v4f lsum = {0.0f, 0.0f, 0.0f, 0.0f};
    vec16 ones;
    {
        const int rowi = lane & 15, kgrp = lane >> 4;
        const int one = ((rowi == 0 && !(kgrp & 1)) || (rowi == 1 && (kgrp & 1))) ? (int)T::kOnes : 0;
        const v4i w = {one, one, one, one};
        __builtin_memcpy(&ones, &w, 16);
    }
    // V chunk t -> this wave's area: piece pc = rows 4 pc .. 4 pc + 3, lane i copies the 16-byte chunk (i & 15) ^ f(r) of row r = 4 pc + (i >> 4)
    unsigned char* varea = smem + wave * VCH;
    const int vr = lane >> 4, vc = lane & 15;
    auto dma_v = [&](int t) {
#pragma unroll
        for (int pc = 0; pc < VCH / 1024; pc++) {
            const int r = 4 * pc + vr;
            const int f = ((r & 3) << 2) | ((r >> 2) & 3);
            const unsigned char* src = vg16 + (long)min(t * 64 + r, p.Skv - 1) * v16_row_stride(p, RB) + ((vc ^ f) << 4);   // (keys beyond Skv: the last row; their P is 0)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(varea + (pc << 10)), 16, 0, 0);
        }
    };
    // per-lane pieces of the transposed-read addresses (pv16_block_pass, D = 128)
    const int q4 = (lane >> 2) & 3, p4 = lane & 3, cg = (lane >> 4) & 1;
    const int cc = 2 * cg + (p4 >> 1);
    const unsigned tr_lo = (unsigned)RB * (4 * hh + q4) + 16u * (cc ^ hh) + 8u * (p4 & 1);
    const unsigned tr_hi = (unsigned)RB * (4 * hh + 8 + q4) + 16u * (cc ^ (hh + 2)) + 8u * (p4 & 1);
    const unsigned vaddr = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)varea;
    unsigned alo[MB], ahi[MB];
#pragma unroll
    for (int m = 0; m < MB; m++) {
        const unsigned xm = 64u * (unsigned)(m ^ q4);
        alo[m] = vaddr + tr_lo + xm;
        ahi[m] = vaddr + tr_hi + xm;
    }
    for (int t = t0; t < t1; t++) {
        // the chunk's K fragments are requested FIRST: their round trip runs under the sixteen LDS-DMA requests of its V (60 - 100 issue
        // cycles each for the requesting wave), which in turn fly under QK^T and the exponentials
        const unsigned char* kc = kg + (long)t * CH + frag_lane_off;
        v8i kf[2 * KS];
#pragma unroll
        for (int s = 0; s < KS; s++) {
            kf[2 * s] = gload_frag(kc + ((0 * KS + s) << 11));
            kf[2 * s + 1] = gload_frag(kc + ((1 * KS + s) << 11));
        }
        asm volatile("" ::: "memory");
        dma_v(t);
        v16f s0, s1;
#pragma unroll
        for (int r = 0; r < 16; r++) { s0[r] = 0.0f; s1[r] = 0.0f; }
#pragma unroll
        for (int s = 0; s < KS; s++) {
            s0 = mfma_f8<QK_FMT, QK_FMT>(kf[2 * s], qf[s], s0);
            s1 = mfma_f8<QK_FMT, QK_FMT>(kf[2 * s + 1], qf[s], s1);
        }
        prep_scores<CAUSAL, false>(s0, s1, p, t * 64, row_lo, row, hh, nullptr);
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
            lsum[0] *= alpha;
            lsum[1] *= __uint_as_float(swizzle_xor16(__float_as_uint(alpha)));
            m_run = m_new;
        }
        const float mc = -m_run * c;
        vec16 pb[2][2];
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
                }
                const v4i wv = {(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
                __builtin_memcpy(&pb[tt][s], &wv, 16);
                lsum = T::mfma_sum(ones, pb[tt][s], lsum);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // V(t) has landed in this wave's area
        // O^T += V^T.P^T, one k-step (16 keys) after the other over the four column blocks, the next k-step's transposed reads in flight
        {
            v2i32 ra[2 * MB], rb[2 * MB];
#define QATTN_R16_ISSUE(R, J)                                                  \
    _Pragma("unroll") for (int m = 0; m < MB; m++) {                           \
        R[2 * m] = pv16_read_tr_at<RB * 16 * J>(alo[m]);                       \
        R[2 * m + 1] = pv16_read_tr_at<RB * 16 * J>(ahi[m]);                   \
    }
#define QATTN_R16_MUL(R, PJ) _Pragma("unroll") for (int m = 0; m < MB; m++) o[m] = T::mfma(pv16_operand<vec16>(R[2 * m], R[2 * m + 1]), PJ, o[m]);
            QATTN_R16_ISSUE(ra, 0)
            QATTN_R16_ISSUE(rb, 1)
            pv16_wait_lds<2 * MB>(ra);
            QATTN_R16_MUL(ra, pb[0][0])
            QATTN_R16_ISSUE(ra, 2)
            pv16_wait_lds<2 * MB>(rb);
            QATTN_R16_MUL(rb, pb[0][1])
            QATTN_R16_ISSUE(rb, 3)
            pv16_wait_lds<2 * MB>(ra);
            QATTN_R16_MUL(ra, pb[1][0])
            pv16_wait_lds<0>(rb);   // (every read of the area is complete: the next chunk may overwrite it)
            QATTN_R16_MUL(rb, pb[1][1])
#undef QATTN_R16_ISSUE
#undef QATTN_R16_MUL
        }
    }
    // the row sums, per lane (both half-waves' keys are in them): query q's sits in lane q & 15, register q >> 4
    float l_run;
    {
        const float l_lo = bcast_low16(lsum[0]), l_hi = bcast_low16(lsum[1]);
        l_run = (lane & 16) ? l_hi : l_lo;
    }
    // ---- merge the NW partials pairwise through LDS: {4..7} -> {0..3}, {2,3} -> {0,1}, {1} -> {0}  (rescue_rows_at)
    __syncthreads();   // every wave is done with its V area, which the slots alias
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
        store_o_rows<MB>(p.out, p.out_fmt, o, 1.0f / l_run, out_row_offset(p, bh, row, MB * 64), hh, store && row < p.Sq);
        if (p.lse && hh == 0 && store && row < p.Sq) p.lse[bh * p.lse_stride + row] = (0.6931471805599453f * (m_run * c) + __logf(l_run)) * p.lse_mul;
        if (p.path && hh == 0 && store && row < p.Sq) p.path[bh * p.Sq + row] = (unsigned char)QATTN_PATH_V16;
    }
}

// The 16-bit-V form of qattn_fp8_attention_forward (v_fmt = QATTN_FMT_BF16 / _FP16): every query block through pv16_block_pass, one
// workgroup per block (map_block: XCD-contiguous heads, causal blocks heaviest first).
// n_blocks > 0: only the first n_blocks query blocks of every head (the fused step's early rows on the paths whose main kernel has no
// 16-bit-V pass of its own: token-wise scales, fp16 inputs).
int launch_attn_pv16_dense(const AttnParams& p, int D, int qk_fmt, int v16_fmt, int causal, int scale_mode, hipStream_t st, int n_blocks);
int launch_attn_pv16_sv(const AttnParams& p, int D, int qk_fmt, int v16_fmt, int causal, int scale_mode, hipStream_t st, int n_blocks);   // q / v16 / out through strides
inline int launch_attn_pv16(const AttnParams& p, int D, int qk_fmt, int v16_fmt, int causal, int scale_mode, hipStream_t st, int n_blocks = 0) {
    return attn_params_dense(p, D) ? launch_attn_pv16_dense(p, D, qk_fmt, v16_fmt, causal, scale_mode, st, n_blocks)
                                   : launch_attn_pv16_sv(p, D, qk_fmt, v16_fmt, causal, scale_mode, st, n_blocks);
}
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
