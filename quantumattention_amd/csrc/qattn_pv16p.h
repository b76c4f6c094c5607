// qattn_pv16p.h -- the 16-bit-V pass of the fused D = 128 step, software-pipelined like the fp8 sweep (round 5).
//
// Same numerics as pv16_block_pass (qattn_pv16.h: FP8 QK^T, exact exponentials, 16-bit P cast pairwise, the ORIGINAL 16-bit V through
// ds_read_b64_tr_b16, row sums of the rounded P on the matrix pipe -- the reference kernel's own P.V, src/quantum_attn/tk/attention.py:72,
// 286,318), same LDS image of a V chunk; what differs is the order of the work inside a wave:
//
//  * three-deep pipeline per wave as in qattn_attn_v2.hip full_step: iteration t issues PV(t-2), the row sums of P(t-2) and QK^T(t) while
//    the vector pipe runs the softmax of chunk t-1 -- nothing in an iteration waits for a product of the same iteration;
//  * seven slots of 128 matrix-pipe cycles: PV k-steps 0..3 (four 32x32x16 products each, one per 32-column block of O^T), the four
//    row-sum products, QK^T on K fragments (tile 0 and 1, k-step 0), QK^T (k-step 1).  Every operand group is requested one slot ahead
//    into one of TWO 16-register sets that serve the transposed V reads and the K fragments alike; all LDS reads are asm with counted
//    waits (behind a builtin LDS load the compiler drains every LDS-DMA in flight, and it cannot count reads it does not know);
//  * ring of 5 stages {K(t) 8 KiB | V16(t-1) 16 KiB} filled by LDS-DMA (three 1 KiB pieces per wave and stage), the waves meet every
//    second iteration (qattn_attn_v2.hip kv_sweep's protocol).
// Un-pipelined (pv16_block_pass: QK^T -> softmax -> PV per wave in turn, one barrier per chunk) the pass spent 2930 cycles per chunk on
// 1664 of products.
#pragma once
#include "qattn_pv16.h"

namespace qattn {

constexpr int kP16Stages = 5;
constexpr int kP16Sync = 2;

__device__ __forceinline__ v4i p16_read_b128(unsigned addr) {
    v4i r;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
template <int OFF>
__device__ __forceinline__ v4i p16_read_b128_at(unsigned addr) {
    v4i r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
// Two operand sets of 16 registers each: the eight transposed reads of one PV k-step (two per 32-column block) or two K fragments (four
// 16-byte reads).  The registers an asm load writes may only be touched behind the wait that covers it -- the compiler takes an asm
// output for ready at once, and a copy it places between the load and the wait reads whatever the register held before (found on
// the GPU: waves lost the race by a few cycles and multiplied stale K fragments) -- so the waits take the loads' OWN outputs as operands,
// and nothing re-packs them in between.
struct P16Set {
    v2i32 r[8];   // transposed V reads: {r[2 m], r[2 m + 1]} = the A operand of row block m
    v4i k[4];     // K fragments: {k[0], k[1]} and {k[2], k[3]}
};
template <int N>
__device__ __forceinline__ void p16_wait(P16Set& a) {   // ... for the set's transposed reads
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(a.r[0]), "+v"(a.r[1]), "+v"(a.r[2]), "+v"(a.r[3]), "+v"(a.r[4]), "+v"(a.r[5]), "+v"(a.r[6]), "+v"(a.r[7])
                 : "n"(N)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void p16_wait_k(P16Set& a) {   // ... for the set's K fragments
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a.k[0]), "+v"(a.k[1]), "+v"(a.k[2]), "+v"(a.k[3]) : "n"(N) : "memory");
}
__device__ __forceinline__ v8i p16_kfrag(const P16Set& a, int which) {   // fragment `which` (0 / 1) of a set filled by p16_read_k
    const v4i lo = a.k[2 * which], hi = a.k[2 * which + 1];
    return v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// two K fragments (32 bytes per lane each: pieces 512 bytes apart) at kaddr + OFF0 / OFF1
template <int OFF0, int OFF1>
__device__ __forceinline__ void p16_read_k(P16Set& a, unsigned kaddr) {
    a.k[0] = p16_read_b128_at<OFF0>(kaddr);
    a.k[1] = p16_read_b128_at<OFF0 + 512>(kaddr);
    a.k[2] = p16_read_b128_at<OFF1>(kaddr);
    a.k[3] = p16_read_b128_at<OFF1 + 512>(kaddr);
}

template <int V16_FMT>
struct Wave16 {
    typedef typename Pv16Type<V16_FMT>::vec vec16;
    v16f o[4];          // O^T accumulators
    v16f s[2][2];       // S^T ping-pong
    vec16 p[2][4];      // P ping-pong: p[t & 1][j] = the B operand of PV k-step j (keys 16 j .. 16 j + 15 of the chunk)
    v8i qf[2];          // Q^T fragments
    v4f lsum;
    vec16 ones;
    float m_run, c, mcv, lim;
};

#define P16_FENCE() __builtin_amdgcn_sched_barrier(0)

// One pipelined iteration (1 <= t <= n_w).  PAR = t & 1.
//   kaddr : LDS address of stage(t)'s K part + the lane's fragment offset        vlo / vhi : the lane's transposed-read addresses (row
//   blocks 0..3) into stage(t-1)'s V part = V(t-2);  vnlo / vnhi: the same into stage(t)'s V part = V(t-1), for the next iteration's
//   first k-step.  A holds the transposed reads of k-step 0 on entry (requested by the previous iteration) and on exit.
template <int QK_FMT, int V16_FMT, int PAR>
__device__ __forceinline__ void p16_step(Wave16<V16_FMT>& st, P16Set& A, P16Set& B, unsigned kaddr, const unsigned (&vlo)[4], const unsigned (&vhi)[4],
                                         const unsigned (&vnlo)[4], const unsigned (&vnhi)[4]) {
    typedef Pv16Type<V16_FMT> T;
    typedef typename T::vec vec16;
    constexpr int RB = 256;
    const v16f (&sc)[2] = st.s[PAR ^ 1];
    v16f (&sn)[2] = st.s[PAR];
    vec16 (&pc)[4] = st.p[PAR ^ 1];
    const vec16 (&pp)[4] = st.p[PAR];
    const float c = st.c, mc = st.mcv;
    // softmax pieces: pair group g (0 .. 7) = registers 4 g .. 4 g + 3 of the chunk's 32 scores -> two dwords of P k-step g >> 1
    unsigned pw[16];
    auto exp4 = [&](int g) __attribute__((always_inline)) {
        const v16f& sx = sc[g >> 2];
        const int b = 4 * (g & 3);
        const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[b], c, mc)), e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[b + 1], c, mc));
        const float e2 = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[b + 2], c, mc)), e3 = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[b + 3], c, mc));
        pw[2 * g] = T::pack2(e0, e1);
        pw[2 * g + 1] = T::pack2(e2, e3);
    };
    auto trk = [&](P16Set& X, auto j_tag, const unsigned (&lo)[4], const unsigned (&hi)[4]) __attribute__((always_inline)) {
        constexpr int J = decltype(j_tag)::value;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            X.r[2 * m] = pv16_read_tr_at<RB * 16 * J>(lo[m]);
            X.r[2 * m + 1] = pv16_read_tr_at<RB * 16 * J>(hi[m]);
        }
    };
    auto pvk = [&](P16Set& X, const vec16& pj) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 4; m++) st.o[m] = T::mfma(pv16_operand<vec16>(X.r[2 * m], X.r[2 * m + 1]), pj, st.o[m]);
    };
    using J1 = std::integral_constant<int, 1>;
    using J2 = std::integral_constant<int, 2>;
    using J3 = std::integral_constant<int, 3>;
    using J0 = std::integral_constant<int, 0>;
    // slot 0: PV k-step 0 (operands in A)            requests: k-step 1 -> B            VALU: maxima of tile 0
    p16_wait<0>(A);
    pvk(A, pp[0]);
    P16_FENCE();
    trk(B, J1{}, vlo, vhi);
    float mxa = max3_raw(sc[0][0], sc[0][1], sc[0][2]), mxb = max3_raw(sc[0][3], sc[0][4], sc[0][5]), mxc = max3_raw(sc[0][6], sc[0][7], sc[0][8]);
    mxa = max3_raw(mxa, sc[0][9], sc[0][10]); mxb = max3_raw(mxb, sc[0][11], sc[0][12]); mxc = max3_raw(mxc, sc[0][13], sc[0][14]);
    mxa = max3_raw(mxa, sc[0][15], sc[1][0]); mxb = max3_raw(mxb, sc[1][1], sc[1][2]); mxc = max3_raw(mxc, sc[1][3], sc[1][4]);
    exp4(0);
    P16_FENCE();
    // slot 1: PV k-step 1 (B)                        requests: k-step 2 -> A            VALU: maxima of tile 1, groups 1, 2
    p16_wait<0>(B);
    pvk(B, pp[1]);
    P16_FENCE();
    trk(A, J2{}, vlo, vhi);
    mxa = max3_raw(mxa, sc[1][5], sc[1][6]); mxb = max3_raw(mxb, sc[1][7], sc[1][8]); mxc = max3_raw(mxc, sc[1][9], sc[1][10]);
    mxa = max3_raw(mxa, sc[1][11], sc[1][12]); mxb = max3_raw(mxb, sc[1][13], sc[1][14]); mxc = max3_raw(mxc, sc[1][15], sc[1][15]);
    float mx = max3_raw(mxa, mxb, mxc);
    exp4(1);
    exp4(2);
    P16_FENCE();
    // slot 2: PV k-step 2 (A)                        requests: k-step 3 -> B            VALU: groups 3, 4
    p16_wait<0>(A);
    pvk(A, pp[2]);
    P16_FENCE();
    trk(B, J3{}, vlo, vhi);
    exp4(3);
    exp4(4);
    P16_FENCE();
    // slot 3: PV k-step 3 (B)                        requests: K(tile 0, k-step 0), K(tile 1, k-step 0) -> A      VALU: group 5
    p16_wait<0>(B);
    pvk(B, pp[3]);
    P16_FENCE();
    p16_read_k<(0 << 11), (2 << 11)>(A, kaddr);
    exp4(5);
    {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = max3_raw(__uint_as_float(sw[0]), __uint_as_float(sw[1]), __uint_as_float(sw[1]));
    }
    P16_FENCE();
    // slot 4: row sums of the rounded P(t-2)         requests: K(tile 0, k-step 1), K(tile 1, k-step 1) -> B      VALU: group 6
#pragma unroll
    for (int j = 0; j < 4; j++) st.lsum = T::mfma_sum(st.ones, pp[j], st.lsum);
    P16_FENCE();
    p16_read_k<(1 << 11), (3 << 11)>(B, kaddr);
    exp4(6);
    P16_FENCE();
    // slot 5: S(t) = K.Q^T, k-step 0 (A)                                                 VALU: group 7
    p16_wait_k<4>(A);
#pragma unroll
    for (int r = 0; r < 16; r++) { sn[0][r] = 0.0f; sn[1][r] = 0.0f; }
    sn[0] = mfma_f8<QK_FMT, QK_FMT>(p16_kfrag(A, 0), st.qf[0], sn[0]);
    sn[1] = mfma_f8<QK_FMT, QK_FMT>(p16_kfrag(A, 1), st.qf[0], sn[1]);
    P16_FENCE();
    exp4(7);
    P16_FENCE();
    // slot 6: S(t) += K.Q^T, k-step 1 (B)            requests: next iteration's k-step 0 -> A
    p16_wait_k<0>(B);
    sn[0] = mfma_f8<QK_FMT, QK_FMT>(p16_kfrag(B, 0), st.qf[1], sn[0]);
    sn[1] = mfma_f8<QK_FMT, QK_FMT>(p16_kfrag(B, 1), st.qf[1], sn[1]);
    P16_FENCE();
    trk(A, J0{}, vnlo, vnhi);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const v4i wv = {(int)pw[4 * j], (int)pw[4 * j + 1], (int)pw[4 * j + 2], (int)pw[4 * j + 3]};
        __builtin_memcpy(&pc[j], &wv, 16);
    }
    P16_FENCE();
    // rare fix-up: a row's maximum grew beyond the deferred-rescale threshold -- rescale O and the row sums (they include chunk t-2) and
    // redo this chunk's exponentials against the new reference
    if (__builtin_expect(__any(mx > st.lim) != 0, 0)) {
        const float m_new = fmaxf(st.m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((st.m_run - m_new) * c);
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) st.o[m][r] *= alpha;
        st.lsum[0] *= alpha;
        st.lsum[1] *= __uint_as_float(swizzle_xor16(__float_as_uint(alpha)));
        st.m_run = m_new;
        const float mc2 = -m_new * c;
        st.mcv = mc2;
        st.lim = m_new + kPv16RescaleThr / c;
#pragma unroll
        for (int g = 0; g < 8; g++) {
            const v16f& sx = sc[g >> 2];
            const int b = 4 * (g & 3);
            const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[b], c, mc2)), e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[b + 1], c, mc2));
            const float e2 = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[b + 2], c, mc2)), e3 = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[b + 3], c, mc2));
            pw[2 * g] = T::pack2(e0, e1);
            pw[2 * g + 1] = T::pack2(e2, e3);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const v4i wv = {(int)pw[4 * j], (int)pw[4 * j + 1], (int)pw[4 * j + 2], (int)pw[4 * j + 3]};
            __builtin_memcpy(&pc[j], &wv, 16);
        }
    }
}

// One 256-row query block of the fused step (bf16 / fp16 q, k, v; Q quantised here with the pre-pass's quant8 sequence).
template <int D, int NW, int QK_FMT, int V16_FMT, bool CAUSAL, typename DrawIssue, typename DrawFinish>
__device__ __forceinline__ void pv16p_block_pass(const AttnParams& p, unsigned char* smem, int tid, int bid, DrawIssue&& draw_issue_hook, DrawFinish&& draw_finish_hook) {
    static_assert(D == 128 && NW == 8, "hand-placed slots for D = 128, one K and two V pieces per wave and stage");
    typedef Pv16Type<V16_FMT> T;
    typedef typename T::vec vec16;
    constexpr int CH = 64 * D, RB = 2 * D, VCH = 64 * RB, STAGE = CH + VCH, KS = 2, MB = 4;
    static_assert(kP16Stages * STAGE <= 160 * 1024 - 4096, "the ring fits the CU's LDS");
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
    const unsigned char* vg = v16_head(p, b, h / (p.Hq / p.Hkv), kv_head, RB);
    const long vrs = v16_row_stride(p, RB);   // (a strided view: rows vrs bytes apart)
    const int n_wg = CAUSAL ? min(p.nchunks, (min(q0_wg + QWG, p.Sq) - 1) / 64 + 1) : p.nchunks;
    const int n_w = CAUSAL ? min(n_wg, (q0 + kQPerWave - 1) / 64 + 1) : p.nchunks;
    const int T_ = n_wg + 2;   // iterations t = 0 .. n_wg + 1

    // ---- stage(t) = {K(min(t, n - 1)), V(min(max(t - 1, 0), n - 1))} by LDS-DMA: piece w of K, pieces 2 w and 2 w + 1 of V (four rows
    // each; lane i copies the 16-byte chunk (i & 15) ^ f(r) of row r: the XOR image of qattn_pv16.h)
    const int vr = lane >> 4, vc = lane & 15;
    unsigned voff_lane[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int r = 4 * (2 * wave + i) + vr;
        const int f = ((r & 3) << 2) | ((r >> 2) & 3);
        voff_lane[i] = (unsigned)r * (unsigned)vrs + (unsigned)((vc ^ f) << 4);
    }
    const unsigned kpiece = ((unsigned)wave << 10) + ((unsigned)lane << 4);
    const int last = p.nchunks - 1;
    int t_next = 0;
    unsigned lds_next = 0;
    auto dma_stage = [&]() __attribute__((always_inline)) {
        const int tk = min(t_next, last), tv = min(max(t_next - 1, 0), last);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kg + (long)tk * CH + kpiece),
                                         (__attribute__((address_space(3))) void*)(smem + lds_next + (wave << 10)), 16, 0, 0);
        const bool ragged = tv * 64 + 64 > p.Skv;   // (workgroup-uniform) the head's last, ragged chunk: keys beyond Skv re-read the last row
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const unsigned char* src = vg + (long)tv * (64 * vrs) + voff_lane[i];
            if (ragged) {
                const int r = 4 * (2 * wave + i) + vr;
                const int f = ((r & 3) << 2) | ((r >> 2) & 3);
                src = vg + (long)min(tv * 64 + r, p.Skv - 1) * vrs + ((vc ^ f) << 4);
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem + lds_next + CH + ((2 * wave + i) << 10)), 16, 0, 0);
        }
        ++t_next;
        lds_next = lds_next + STAGE == kP16Stages * STAGE ? 0u : lds_next + STAGE;
    };
#pragma unroll
    for (int g = 0; g < kP16Sync; g++)
        if (g < T_) dma_stage();

    // ---- Q^T fragments (registers) and the softmax scale
    Wave16<V16_FMT> st;
    {
        const float inv_qmax = (float)(1.0 / (double)(QK_FMT == QATTN_FMT_E4M3 ? 448.0 : 57344.0));
        const float scale_q = make_scale(__uint_as_float(max_partials(p.q_amax_part + bh * p.amax_stride, p.amax_n, lane) & 0x7fffffffu), inv_qmax,
                                         p.q_numerics, V16_FMT);
        if (q0_wg == 0 && tid == 0) p.sq_out[bh] = scale_q;
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
            st.qf[s] = v8i{w[0].x, w[0].y, w[1].x, w[1].y, w[2].x, w[2].y, w[3].x, w[3].y};
        }
        st.c = p.sm_log2e * scale_q * p.sk[kv_head];
    }
#pragma unroll
    for (int m = 0; m < MB; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) st.o[m][r] = 0.0f;
    st.lsum = v4f{0.0f, 0.0f, 0.0f, 0.0f};
    st.m_run = -1.0e30f;
    st.mcv = 0.0f;
    st.lim = -1.0e30f;
    {
        const int rowi = lane & 15, kgrp = lane >> 4;
        const int one = ((rowi == 0 && !(kgrp & 1)) || (rowi == 1 && (kgrp & 1))) ? (int)T::kOnes : 0;
        const v4i w = {one, one, one, one};
        __builtin_memcpy(&st.ones, &w, 16);
        const v4i z = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; j++) { __builtin_memcpy(&st.p[0][j], &z, 16); __builtin_memcpy(&st.p[1][j], &z, 16); }
    }
    // ---- per-lane LDS addresses: the K fragment offset; the transposed-read bases (pv16_block_pass) with the row block's 64-byte
    // granule m ^ q4 folded in -- relative to a stage's start, the stage offset is added per iteration
    const unsigned smem_a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned kfrag = smem_a + (unsigned)((hh << 10) + (ql << 4));
    const int q4 = (lane >> 2) & 3, p4 = lane & 3, cg = (lane >> 4) & 1;
    const int cc = 2 * cg + (p4 >> 1);
    const unsigned tr_lo = smem_a + CH + (unsigned)RB * (4 * hh + q4) + 16u * (cc ^ hh) + 8u * (p4 & 1);
    const unsigned tr_hi = smem_a + CH + (unsigned)RB * (4 * hh + 8 + q4) + 16u * (cc ^ (hh + 2)) + 8u * (p4 & 1);
    unsigned xm[4];
#pragma unroll
    for (int m = 0; m < 4; m++) xm[m] = 64u * (unsigned)(m ^ q4);
    auto vaddr = [&](unsigned slot, unsigned (&lo)[4], unsigned (&hi)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 4; m++) { lo[m] = tr_lo + slot + xm[m]; hi[m] = tr_hi + slot + xm[m]; }
    };
    unsigned slot_cur = 0, slot_prev = 0;
    auto advance = [&]() __attribute__((always_inline)) {
        slot_prev = slot_cur;
        slot_cur = slot_cur + STAGE == kP16Stages * STAGE ? 0u : slot_cur + STAGE;
    };
    auto sync_iter = [&](int t) __attribute__((always_inline)) {
        if (t % kP16Sync == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of stages t .. t + 1 have landed
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int g = 0; g < kP16Sync; g++)
                if (t + kP16Sync + g < T_) dma_stage();
        } else {
            asm volatile("s_nop 0" ::: "memory");
        }
    };
    P16Set A, B;
    // ---- t = 0: QK^T(0); the row's reference starts at chunk 0's maximum
    {
        sync_iter(0);
        const unsigned kaddr = kfrag + slot_cur;
        p16_read_k<(0 << 11), (2 << 11)>(A, kaddr);
        p16_read_k<(1 << 11), (3 << 11)>(B, kaddr);
        p16_wait_k<4>(A);
#pragma unroll
        for (int r = 0; r < 16; r++) { st.s[0][0][r] = 0.0f; st.s[0][1][r] = 0.0f; }
        st.s[0][0] = mfma_f8<QK_FMT, QK_FMT>(p16_kfrag(A, 0), st.qf[0], st.s[0][0]);
        st.s[0][1] = mfma_f8<QK_FMT, QK_FMT>(p16_kfrag(A, 1), st.qf[0], st.s[0][1]);
        p16_wait_k<0>(B);
        st.s[0][0] = mfma_f8<QK_FMT, QK_FMT>(p16_kfrag(B, 0), st.qf[1], st.s[0][0]);
        st.s[0][1] = mfma_f8<QK_FMT, QK_FMT>(p16_kfrag(B, 1), st.qf[1], st.s[0][1]);
        // k-step 0 of stage(0)'s V part (= V(0), multiplied by P = 0 at t = 1)
        unsigned lo[4], hi[4];
        vaddr(slot_cur, lo, hi);
#pragma unroll
        for (int m = 0; m < 4; m++) { A.r[2 * m] = pv16_read_tr_at<0>(lo[m]); A.r[2 * m + 1] = pv16_read_tr_at<0>(hi[m]); }
        advance();
        prep_scores<CAUSAL, false, true>(st.s[0][0], st.s[0][1], p, 0, q0, qrow, hh, nullptr);
        float mx0 = max32_after_mfma(st.s[0][0], st.s[0][1]);
        const auto sw0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx0), __float_as_uint(mx0), false, false);
        mx0 = fmaxf(__uint_as_float(sw0[0]), __uint_as_float(sw0[1]));
        st.m_run = fmaxf(st.m_run, mx0);
        st.mcv = -st.m_run * st.c;
        st.lim = st.m_run + kPv16RescaleThr / st.c;
    }
    // ---- t = 1 .. n_w: pipelined steps, two per trip; the wave's last chunk (causal diagonal, ragged tail) is masked behind a
    // wave-uniform test in front of the step that exponentiates it
    // keys at or beyond Skv, and (causal) keys above the lane's row -> -inf, without compares: 32 compare masks in scalar register pairs
    // inside the sweep's loop make it spill its scalars.  x = (last live key - key) + 1/2 is positive for a live key, negative for a dead
    // one; x * inf = +-inf; min(score, +-inf) keeps the score or makes it -inf -- prep_scores' result bit for bit.  Register r of key
    // tile kt holds key 64 chunk + 32 kt + (r & 3) + 8 (r >> 2) + 4 hh.
    const int mask_from = min(p.Skv >> 6, CAUSAL ? q0 >> 6 : 0x7fffffff);   // first chunk that needs it (wave-uniform)
    const int last_live = CAUSAL ? min(p.Skv - 1, qrow) : p.Skv - 1;
    auto mask_chunk = [&](auto par_tag, int chunk) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_tag)::value ^ 1;   // S(t-1) lives in st.s[PAR ^ 1] of step t
        const float left = (float)(last_live - chunk * 64 - 4 * hh) + 0.5f;
        float inf = __builtin_inff();
        asm volatile("" : "+v"(inf));
#pragma unroll
        for (int kt = 0; kt < 2; kt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float xr = left - (float)(32 * kt + (r & 3) + 8 * (r >> 2));
                st.s[PAR][kt][r] = __builtin_fminf(st.s[PAR][kt][r], xr * inf);
            }
    };
    auto full = [&](auto par_tag, int t) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_tag)::value;
        sync_iter(t);
        const unsigned kaddr = kfrag + slot_cur;
        unsigned lo[4], hi[4], nlo[4], nhi[4];
        vaddr(slot_prev, lo, hi);
        vaddr(slot_cur, nlo, nhi);
        advance();
        if (t - 1 >= mask_from) mask_chunk(par_tag, t - 1);   // the chunk this step exponentiates reaches past the key range or the diagonal
        p16_step<QK_FMT, V16_FMT, PAR>(st, A, B, kaddr, lo, hi, nlo, nhi);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    int t = 1;
    for (; t + 1 <= n_w; t += 2) {
        full(P1{}, t);
        full(P0{}, t + 1);
    }
    if (t <= n_w) {
        full(P1{}, t);
        ++t;
    }
    // ---- t = n_w + 1: the last chunk's PV and row sums (k-step 0's operands are in A)
    {
        sync_iter(t);
        unsigned lo[4], hi[4];
        vaddr(slot_prev, lo, hi);
        auto tail = [&](auto par_tag) {
            constexpr int PAR = decltype(par_tag)::value;
            const vec16 (&pp)[4] = st.p[PAR];
#pragma unroll
            for (int m = 0; m < 4; m++) { B.r[2 * m] = pv16_read_tr_at<RB * 16>(lo[m]); B.r[2 * m + 1] = pv16_read_tr_at<RB * 16>(hi[m]); }
            p16_wait<8>(A);
#pragma unroll
            for (int m = 0; m < 4; m++) st.o[m] = T::mfma(pv16_operand<vec16>(A.r[2 * m], A.r[2 * m + 1]), pp[0], st.o[m]);
#pragma unroll
            for (int m = 0; m < 4; m++) { A.r[2 * m] = pv16_read_tr_at<RB * 32>(lo[m]); A.r[2 * m + 1] = pv16_read_tr_at<RB * 32>(hi[m]); }
            p16_wait<8>(B);
#pragma unroll
            for (int m = 0; m < 4; m++) st.o[m] = T::mfma(pv16_operand<vec16>(B.r[2 * m], B.r[2 * m + 1]), pp[1], st.o[m]);
#pragma unroll
            for (int m = 0; m < 4; m++) { B.r[2 * m] = pv16_read_tr_at<RB * 48>(lo[m]); B.r[2 * m + 1] = pv16_read_tr_at<RB * 48>(hi[m]); }
            p16_wait<8>(A);
#pragma unroll
            for (int m = 0; m < 4; m++) st.o[m] = T::mfma(pv16_operand<vec16>(A.r[2 * m], A.r[2 * m + 1]), pp[2], st.o[m]);
            p16_wait<0>(B);
#pragma unroll
            for (int m = 0; m < 4; m++) st.o[m] = T::mfma(pv16_operand<vec16>(B.r[2 * m], B.r[2 * m + 1]), pp[3], st.o[m]);
#pragma unroll
            for (int j = 0; j < 4; j++) st.lsum = T::mfma_sum(st.ones, pp[j], st.lsum);
        };
        if (t & 1) tail(P1{}); else tail(P0{});
        ++t;
    }
    // causal: waves whose rows end earlier keep the workgroup's barrier / DMA cadence until the last wave is done
    for (; t < T_; ++t) sync_iter(t);

    const float l_lo = bcast_low16(st.lsum[0]), l_hi = bcast_low16(st.lsum[1]);
    const float l_tot = (lane & 16) ? l_hi : l_lo;
    const unsigned ticket = draw_issue_hook();
    store_o_rows<MB>(p.out, p.out_fmt, st.o, 1.0f / l_tot, out_row_offset(p, bh, qrow, MB * 64), hh, qvalid);
    draw_finish_hook(ticket);
    // optional outputs of the fused entry (ABI 7): the log-sum-exp row (the sums are of the ROUNDED 16-bit P, pv16_block_pass) and the row's path
    if (p.lse && hh == 0 && qvalid) p.lse[bh * p.lse_stride + qrow] = (0.6931471805599453f * (st.m_run * st.c) + __logf(l_tot)) * p.lse_mul;
    if (p.path && hh == 0 && qvalid) p.path[bh * p.Sq + qrow] = (unsigned char)QATTN_PATH_V16;
}

}  // namespace qattn
