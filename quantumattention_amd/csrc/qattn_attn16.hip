// qattn_attn16.hip -- 16-bit (bf16 / fp16) fused attention forward for gfx950: the non-fp8 sibling path.
//
// Replaces the non-TK_ATTN_IS_FP8 build of fwd_attend_ker + launcher (src/quantum_attn/tk/attention.py:212,238-240,
// 289-313, 355-647) behind the op quantum_attn::attention_forward (src/quantum_attn/ops.py:17-45).  Same orientation
// as the fp8 kernels: S^T[key][q] = K.Q^T and O^T[d][q] += V^T.P^T on v_mfma_f32_32x32x16_{bf16,f16}; the query sits
// on the lane, so the softmax state is per-lane and the 16-bit-converted P registers are the PV B operand directly
// (accumulator registers 8s..8s+7 of a 32-key tile = k-step s: element j <-> key 16s + 8(j>>2) + 4h + (j&3)).
//
// K and V are re-laid by qattn_pack16 into fragment order (64-key chunks, 16-byte pieces):
//   K16FRAG chunk = [t:2][s:D/16][hh:2][key:32][8 x 16 bit]   piece = K[64c + 32t + key][16s + 8hh + (0..7)]
//   V16FRAG chunk = [t:2][m:D/32][s:2][hh:2][d:32][8 x 16 bit] piece j = V[64c + 32t + 16s + 8(j>>2) + 4hh + (j&3)][32m + d]
// so every 32-key tile (t) of K and of V is one linear LDS-DMA copy and every MFMA A operand is one conflict-free
// ds_read_b128.  Exact exp2, fp32 row sums, deferred rescale; structure described at the kernel.
#include "qattn_attn.h"

namespace qattn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int FMT16> struct T16;
template <> struct T16<QATTN_FMT_BF16> {
    typedef __bf16 elt;
    typedef bf16x8 vec;
    static __device__ __forceinline__ v16f mfma(vec a, vec b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ v4f mfma16(vec a, vec b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static constexpr unsigned kOne2 = 0x3f803f80u;                   // two 1.0 elements
    static constexpr float kBitsPerOctave = 128.0f, kExpBias = 127.0f;  // bits of 2^x = (x + 127) * 2^7 (+ mantissa)
};
template <> struct T16<QATTN_FMT_FP16> {
    typedef _Float16 elt;
    typedef f16x8 vec;
    static __device__ __forceinline__ v16f mfma(vec a, vec b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ v4f mfma16(vec a, vec b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static constexpr unsigned kOne2 = 0x3c003c00u;
    static constexpr float kBitsPerOctave = 1024.0f, kExpBias = 15.0f;
};

struct Attn16Params {
    const unsigned char* q;  // [B,Hq,Sq,D] 16-bit row-major
    const unsigned char* k;  // K16FRAG
    const unsigned char* v;  // V16FRAG
    void* out;
    float* lse;
    int B, Hq, Hkv, Sq, Skv;
    int nqb, ntiles, xcd_remap;
    float sm_log2e;
    int fast_exp;  // opt-in linear-mantissa exponential for rows that see >= kTwoTermKeys keys
    long q_bs, q_hs, q_rs;   // byte strides of q: batch, head, row (a strided view with D innermost; dense: Hq Sq D 2, Sq D 2, D 2)
    long o_bs, o_hs, o_rs;   // and of out
};

constexpr int kWaves16 = 4;                        // 128 query rows per workgroup
constexpr int kQPerWG16 = kWaves16 * kQPerWave;
constexpr int kStages16 = 2;

// Structure (same idea as qattn_attn_v4.hip): every wave runs the plain sequence QK^T -> max / rescale -> exponentials -> PV on
// 32-key tiles with S, P single-buffered and Q^T in registers (<= 168 VGPRs); a workgroup is 4 waves x 32 rows with a
// two-stage ring of {K tile, V tile} (32 KiB for D = 128), so a CU holds three independent workgroups and each SIMD
// interleaves three waves that are never phase-aligned.  One barrier per tile.
// FAST: the 16-bit pattern of 2^x is taken as (x + bias) * 2^mantissa_bits (Schraudolph's linear-mantissa exponential:
// one v_fma_f32 per score + one v_cvt_pknorm_u16_f32 per pair, which already packs the PV B operand; relative error
// 1.8 % rms, -3.9 .. +2.0 %), and the row sums of the SAME approximated weights come from two small
// v_mfma_f32_16x16x32 per tile with a two-row selector A operand, so numerator and denominator stay consistent.  With the
// exact path (v_exp_f32 + fp32 sums) the kernel is VALU-bound.  FAST is OPT-IN (qattn_attention_forward_16's fast_exp
// argument, config.attention.fast_exp16): its per-weight error only averages out over rows whose weight is spread over
// many keys, so it is then used where a row sees >= kTwoTermKeys keys and no LSE is requested; the exact instantiation
// (the reference's numerics: exact exp2, 16-bit P) is the default everywhere.
constexpr float kFastExpBias = -0.0575f;  // centres the (1+f)/2^f mantissa error

template <int D, int FMT16, bool CAUSAL, bool FAST>
__global__ __launch_bounds__(kWaves16 * 64, (D == 256 ? 2 : 3)) void attn16_fwd_kernel(const Attn16Params p, const int qb_lo, const int qb_n) {
    typedef typename T16<FMT16>::vec vec16;
    typedef typename T16<FMT16>::elt elt16;
    constexpr int TB = 32 * D * 2;       // bytes of one 32-key K (or V) tile
    constexpr int STAGE = 2 * TB;
    constexpr int KS = D / 16;           // QK^T k-steps
    constexpr int MB = D / 32;           // O^T row blocks
    constexpr int RK = TB / (kWaves16 * 1024);  // 1 KiB DMA pieces per wave for the K (and the V) tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hh = lane >> 5;
    int bid = blockIdx.x, head, qb;
    if (p.xcd_remap) {
        const int xcd = bid & 7, idx = bid >> 3;
        head = xcd * ((p.B * p.Hq) >> 3) + idx / qb_n;
        qb = idx % qb_n;
    } else {
        head = bid / qb_n;
        qb = bid % qb_n;
    }
    if (CAUSAL) qb = qb_n - 1 - qb;
    qb += qb_lo;
    const int b = head / p.Hq, h = head % p.Hq;
    const long kv_head = (long)b * p.Hkv + h / (p.Hq / p.Hkv);
    const int q0_wg = qb * kQPerWG16, q0 = q0_wg + wave * kQPerWave, qrow = q0 + ql;
    const long head_bytes = (long)ceil_div(p.Skv, 64) * 2 * TB;   // the packed tensors pad S to a multiple of 64 keys
    const unsigned char* kg_w = p.k + kv_head * head_bytes + (wave << 10);
    const unsigned char* vg_w = p.v + kv_head * head_bytes + (wave << 10);
    const int n_wg = CAUSAL ? min(p.ntiles, (min(q0_wg + kQPerWG16, p.Sq) - 1) / 32 + 1) : p.ntiles;
    const int n_w = CAUSAL ? min(n_wg, (q0 + kQPerWave - 1) / 32 + 1) : p.ntiles;

    const unsigned lane16 = (unsigned)lane << 4;
    unsigned toff = 0, slot_next = 0;
    auto dma_next = [&]() {
        unsigned char* dst = smem + slot_next + (wave << 10);
#pragma unroll
        for (int r = 0; r < RK; r++) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kg_w + (toff + lane16 + r * (kWaves16 * 1024))),
                                             (__attribute__((address_space(3))) void*)(dst + r * (kWaves16 * 1024)), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vg_w + (toff + lane16 + r * (kWaves16 * 1024))),
                                             (__attribute__((address_space(3))) void*)(dst + TB + r * (kWaves16 * 1024)), 16, 0, 0);
        }
        toff += TB;
        slot_next ^= STAGE;
    };
    dma_next();

    // Q^T fragments: lane (q, hh) holds Q[q][16s + 8hh + (0..7)] for every k-step s
    vec16 qf[KS];
    {
        const bool qvalid = qrow < p.Sq;
        const unsigned char* qp = p.q + (long)b * p.q_bs + (long)h * p.q_hs + (long)(qvalid ? qrow : 0) * p.q_rs + hh * 16;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            v4i raw = *reinterpret_cast<const v4i*>(qp + s * 32);
            if (!qvalid) raw = v4i{0, 0, 0, 0};
            __builtin_memcpy(&qf[s], &raw, 16);
        }
    }
    const float c = p.sm_log2e;
    v16f o[MB];
#pragma unroll
    for (int m = 0; m < MB; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[m][r] = 0.0f;
    float m_run = -1.0e30f, l_run = 0.0f;
    const int frag_lane_off = (hh << 9) + (ql << 4);
    v4f lsum = {0.0f, 0.0f, 0.0f, 0.0f};
    vec16 ones;  // FAST: A of the row-sum MFMA: lane = row (l & 15) + 16 * k-group; rows 0 / 1 are 1.0 on even / odd k-groups
    {
        const int row = lane & 15, kg = lane >> 4;
        const unsigned one = ((row == 0 && !(kg & 1)) || (row == 1 && (kg & 1))) ? T16<FMT16>::kOne2 : 0u;
        const unsigned w4[4] = {one, one, one, one};
        __builtin_memcpy(&ones, w4, 16);
    }
    constexpr float U16 = 1.0f / 65535.0f, BPO = T16<FMT16>::kBitsPerOctave;
    const float cfast = (BPO * U16) * c;

    for (int t = 0; t < n_wg; t++) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (t + 1 < n_wg) dma_next();
        if (t >= n_w) continue;  // causal: this wave's rows end before tile t
        const unsigned char* kbuf = smem + (t & 1) * STAGE + frag_lane_off;
        const unsigned char* vbuf = kbuf + TB;
        // ---- S^T (32 keys x 32 queries) = K.Q^T
        v16f sc;
#pragma unroll
        for (int r = 0; r < 16; r++) sc[r] = 0.0f;
        // all K fragments of the tile are requested before the MFMA chain starts (left alone the compiler re-uses one
        // register quad and waits for every ds_read right before the MFMA that consumes it)
        // (D = 256: in batches of four, O^T and Q^T alone hold 192 registers)
        constexpr int KB = KS > 8 ? 4 : KS;
#pragma unroll
        for (int s0 = 0; s0 < KS; s0 += KB) {
            vec16 ka[KB];
#pragma unroll
            for (int s = 0; s < KB; s++) ka[s] = *reinterpret_cast<const vec16*>(kbuf + ((s0 + s) << 10));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < KB; s++) sc = T16<FMT16>::mfma(ka[s], qf[s0 + s], sc);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the first batch of V fragments travels while the softmax runs
        constexpr int NV = 2 * MB, VB = NV < 4 ? NV : 4;
        vec16 va[VB];
#pragma unroll
        for (int i = 0; i < VB; i++) va[i] = *reinterpret_cast<const vec16*>(vbuf + (i << 10));
        __builtin_amdgcn_sched_barrier(0);
        const int k0 = t * 32;
        const bool need_mask = (k0 + 32 > p.Skv) || (CAUSAL && k0 + 31 > q0);
        if (__builtin_expect(need_mask, 0)) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const bool dead = key >= p.Skv || (CAUSAL && key > qrow);
                sc[r] = dead ? -INFINITY : sc[r];
            }
        }
        float mx = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, sc[r]), sc[r + 1]);
        mx = fmaxf(mx, sc[15]);
        {
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        if (__builtin_expect(__any((mx - m_run) * c > kRescaleThr) != 0, 0)) {  // deferred rescale (always on the first tile)
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
#pragma unroll
            for (int m = 0; m < MB; m++)
#pragma unroll
                for (int r = 0; r < 16; r++) o[m][r] *= alpha;
            if (FAST) {
                lsum[0] *= alpha;
                lsum[1] *= __shfl(alpha, (lane & 15) + 16);
            } else {
                l_run *= alpha;
            }
            m_run = m_new;
        }
        vec16 pb[2];
        if (FAST) {
            // ---- P bit patterns straight from the scores; row sums on the matrix pipe
            const float off = __builtin_fmaf((-BPO * U16) * m_run, c, (BPO * (T16<FMT16>::kExpBias + kFastExpBias)) * U16);
#pragma unroll
            for (int i = 0; i < 2; i++) {
                unsigned w4[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                    const us2 pk = __builtin_amdgcn_cvt_pknorm_u16(__builtin_fmaf(sc[8 * i + 2 * j], cfast, off),
                                                                   __builtin_fmaf(sc[8 * i + 2 * j + 1], cfast, off));
                    __builtin_memcpy(&w4[j], &pk, 4);
                }
                __builtin_memcpy(&pb[i], w4, 16);
            }
            lsum = T16<FMT16>::mfma16(ones, pb[0], lsum);
            lsum = T16<FMT16>::mfma16(ones, pb[1], lsum);
        } else {
            // ---- P = exp2(c*s - c*m) in fp32, row sums in fp32, 16-bit conversion -> the two PV k-steps' B operands
            const float mc = -m_run * c;
            float ls = 0.0f;
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[8 * i + j], c, mc));
                    ls += e;
                    pb[i][j] = (elt16)e;
                }
            l_run += ls;
        }
        // ---- O^T += V^T.P^T, V fragments in batches of VB: batch b+1 is requested before batch b's MFMAs are issued
#pragma unroll
        for (int b0 = 0; b0 < NV; b0 += VB) {
            vec16 vn[VB];
            __builtin_amdgcn_sched_barrier(0);
            if (b0 + VB < NV) {
#pragma unroll
                for (int i = 0; i < VB; i++) vn[i] = *reinterpret_cast<const vec16*>(vbuf + ((b0 + VB + i) << 10));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < VB; i++) o[(b0 + i) >> 1] = T16<FMT16>::mfma(va[i], pb[(b0 + i) & 1], o[(b0 + i) >> 1]);
            if (b0 + VB < NV) {
#pragma unroll
                for (int i = 0; i < VB; i++) va[i] = vn[i];
            }
        }
    }

    float l_tot;
    if (FAST) {
        const float s0l = __shfl(lsum[0], lane & 15), s1l = __shfl(lsum[1], lane & 15);
        l_tot = (lane & 16) ? s1l : s0l;
    } else {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
        l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const float inv = 1.0f / l_tot;
    if (qrow < p.Sq) {
        elt16* op = reinterpret_cast<elt16*>(reinterpret_cast<unsigned char*>(p.out) + (long)b * p.o_bs + (long)h * p.o_hs + (long)qrow * p.o_rs);
#pragma unroll
        for (int m = 0; m < MB; m++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                typedef elt16 e4 __attribute__((ext_vector_type(4)));
                e4 tv;
#pragma unroll
                for (int i = 0; i < 4; i++) tv[i] = (elt16)(o[m][4 * j + i] * inv);
                *reinterpret_cast<e4*>(op + 32 * m + 8 * j + 4 * hh) = tv;
            }
        if (p.lse && hh == 0) p.lse[((long)b * p.Hq + h) * p.Sq + qrow] = 0.6931471805599453f * (m_run * c) + __logf(l_tot);
    }
}

// 16-bit row-major [G, S, D] -> K16FRAG / V16FRAG.  grid = (ceil(S/64), G), block = 256.
template <int D, int LAYOUT>
__global__ __launch_bounds__(256) void pack16_tile_kernel(const uint4* __restrict__ x, uint4* __restrict__ out, int S, int H, long sb, long sh,
                                                          long ss) {   // sb, sh, ss: 16-byte vectors between batches, heads, rows of x
    constexpr int VPR = D / 8;            // 16-byte vectors (8 elements) per row
    constexpr int RSTRIDE = D * 2 + 4;    // V staging: row stride in bytes (breaks the power-of-two stride for the gather)
    __shared__ __attribute__((aligned(16))) unsigned char img[64 * RSTRIDE];
    const int g = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x, row0 = tile * 64;
    const uint4* xg = x + (long)(g / H) * sb + (long)(g % H) * sh;
    const long Sp = (long)((S + 63) / 64) * 64;
    uint4* og = out + ((long)g * Sp + row0) * VPR;   // a chunk is 64*D*2 bytes = 64*VPR vectors
    if (LAYOUT == QATTN_LAYOUT_K16FRAG) {
        // piece (t, s, hh, key) = K[32t + key][16s + 8hh .. +7] is one 16-byte vector of the source row: pure re-indexing
        for (int i = tid; i < 64 * VPR; i += 256) {
            const int key = i & 31, hh2 = (i >> 5) & 1, s = (i >> 6) % (D / 16), t = i / (64 * (D / 16));
            const int row = row0 + 32 * t + key;
            uint4 raw = make_uint4(0, 0, 0, 0);
            if (row < S) raw = xg[(long)row * ss + 2 * s + hh2];
            og[i] = raw;
        }
    } else {
        for (int vecn = tid; vecn < 64 * VPR; vecn += 256) {
            const int r = vecn / VPR, dv = vecn % VPR, row = row0 + r;
            uint4 raw = make_uint4(0, 0, 0, 0);
            if (row < S) raw = xg[(long)row * ss + dv];
            unsigned* dst = reinterpret_cast<unsigned*>(img + r * RSTRIDE + dv * 16);
            dst[0] = raw.x; dst[1] = raw.y; dst[2] = raw.z; dst[3] = raw.w;
        }
        __syncthreads();
        // output vector n: [t:2][m:D/32][s:2][hh:2][d:32]; its element j = V[32t + 16s + 8(j>>2) + 4hh + (j&3)][32m + d]
        for (int n = tid; n < 64 * VPR; n += 256) {
            const int dl = n & 31, hh2 = (n >> 5) & 1, s = (n >> 6) & 1, m = (n >> 7) % (D / 32), t = n / (128 * (D / 32));
            unsigned short e[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int key = 32 * t + 16 * s + 8 * (j >> 2) + 4 * hh2 + (j & 3);
                e[j] = *reinterpret_cast<const unsigned short*>(img + key * RSTRIDE + (32 * m + dl) * 2);
            }
            uint4 o4;
            __builtin_memcpy(&o4, e, 16);
            og[n] = o4;
        }
    }
}

template <int D, int FMT16, bool CAUSAL, bool FAST>
static int launch16_one(const Attn16Params& p, int row_lo, int row_hi, hipStream_t st) {
    const int qb_lo = row_lo / kQPerWG16, qb_n = ceil_div(min(row_hi, p.Sq), kQPerWG16) - qb_lo;
    if (qb_n <= 0) return QATTN_OK;
    const int grid = p.B * p.Hq * qb_n;
    const size_t lds = (size_t)kStages16 * 2 * 32 * D * 2;
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)attn16_fwd_kernel<D, FMT16, CAUSAL, FAST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return QATTN_ERR_LAUNCH;
    hipLaunchKernelGGL((attn16_fwd_kernel<D, FMT16, CAUSAL, FAST>), dim3(grid), dim3(kWaves16 * 64), lds, st, p, qb_lo, qb_n);
    return QATTN_OK;
}

template <int D, int FMT16>
static int launch16(const Attn16Params& p, int causal, hipStream_t st) {
    // the exact instantiation unless the caller opted into the fast exponential; even then rows that see fewer than
    // kTwoTermKeys keys, and every row when an LSE output is requested, stay exact
    int rows_exact;
    if (!p.fast_exp || p.lse != nullptr) rows_exact = p.Sq;
    else if (causal) rows_exact = min(p.Sq, ceil_div(min(kTwoTermKeys, p.Skv), kQPerWG16) * kQPerWG16);
    else rows_exact = p.Skv < kTwoTermKeys ? p.Sq : 0;
    int rc = QATTN_OK;
    if (rows_exact < p.Sq)
        rc = causal ? launch16_one<D, FMT16, true, true>(p, rows_exact, p.Sq, st) : launch16_one<D, FMT16, false, true>(p, rows_exact, p.Sq, st);
    if (rc == QATTN_OK && rows_exact > 0)
        rc = causal ? launch16_one<D, FMT16, true, false>(p, 0, rows_exact, st) : launch16_one<D, FMT16, false, false>(p, 0, rows_exact, st);
    return rc;
}

}  // namespace qattn

using namespace qattn;

extern "C" size_t qattn_16bit_tensor_bytes(int layout, int B, int H, int S, int D) {
    if (B <= 0 || H <= 0 || S <= 0 || D <= 0) return 0;
    const size_t Sp = layout == QATTN_LAYOUT_ROWMAJOR ? (size_t)S : (size_t)((S + 63) / 64) * 64;
    return (size_t)B * H * Sp * D * 2;
}

// element strides {batch, head, row} of a 16-bit [B,H,S,D] view with D innermost and dense (include/qattn_strided.h); nullptr: dense
static bool strides16_ok(const void* base, const long long* st, int D) {
    if (!st) return true;
    if ((reinterpret_cast<uintptr_t>(base) & 15u) != 0) return false;
    for (int i = 0; i < 3; i++)
        if (st[i] < 0 || st[i] % 8 != 0) return false;
    return st[2] >= D && st[2] <= (1LL << 23);
}

extern "C" int qattn_pack16_strided(const void* x, const long long* strides, void* x_packed, int B, int H, int S, int D, int out_layout,
                                    void* stream) {
    if (!x || !x_packed || B <= 0 || H <= 0 || S <= 0) return QATTN_ERR_INVALID_ARG;
    if (D != 64 && D != 128 && D != 256) return QATTN_ERR_UNSUPPORTED_DIM;
    if (out_layout != QATTN_LAYOUT_K16FRAG && out_layout != QATTN_LAYOUT_V16FRAG) return QATTN_ERR_INVALID_ARG;
    if (!strides16_ok(x, strides, D)) return QATTN_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((S + 63) / 64, B * H), block(256);
    const uint4* xi = (const uint4*)x;
    uint4* xo = (uint4*)x_packed;
    const long vd = D / 8;
    const long sb = strides ? strides[0] / 8 : (long)H * S * vd, sh = strides ? strides[1] / 8 : (long)S * vd, ss = strides ? strides[2] / 8 : vd;
#define PK16(DD, LAY) hipLaunchKernelGGL((pack16_tile_kernel<DD, LAY>), grid, block, 0, st, xi, xo, S, H, sb, sh, ss)
    if (out_layout == QATTN_LAYOUT_K16FRAG) { if (D == 64) PK16(64, QATTN_LAYOUT_K16FRAG); else if (D == 128) PK16(128, QATTN_LAYOUT_K16FRAG); else PK16(256, QATTN_LAYOUT_K16FRAG); }
    else { if (D == 64) PK16(64, QATTN_LAYOUT_V16FRAG); else if (D == 128) PK16(128, QATTN_LAYOUT_V16FRAG); else PK16(256, QATTN_LAYOUT_V16FRAG); }
#undef PK16
    return hipGetLastError() == hipSuccess ? QATTN_OK : QATTN_ERR_LAUNCH;
}

extern "C" int qattn_pack16(const void* x_rowmajor, void* x_packed, int B, int H, int S, int D, int out_layout, void* stream) {
    return qattn_pack16_strided(x_rowmajor, nullptr, x_packed, B, H, S, D, out_layout, stream);
}

extern "C" int qattn_attention_forward_16_strided(const void* q, const long long* strides, const void* k16, const void* v16, void* out, float* lse,
                                                  int B, int Hq, int Hkv, int Sq, int Skv, int D, int fmt, int is_causal, float sm_scale,
                                                  int fast_exp, void* stream) {
    if (!q || !k16 || !v16 || !out) return QATTN_ERR_INVALID_ARG;
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || Sq <= 0 || Skv <= 0) return QATTN_ERR_INVALID_ARG;
    if (D != 64 && D != 128 && D != 256) return QATTN_ERR_UNSUPPORTED_DIM;
    if (Hq % Hkv != 0) return QATTN_ERR_UNSUPPORTED_DIM;
    if (fmt != QATTN_FMT_BF16 && fmt != QATTN_FMT_FP16) return QATTN_ERR_UNSUPPORTED_FMT;
    if (!strides16_ok(q, strides, D) || !strides16_ok(out, strides ? strides + 3 : nullptr, D)) return QATTN_ERR_INVALID_ARG;   // strides: {batch, head, row} of q, then of out
    if (strides && ((B > 1 && strides[3] == 0) || (Hq > 1 && strides[4] == 0))) return QATTN_ERR_INVALID_ARG;   // (`out` cannot be a broadcast view)
    Attn16Params p;
    p.q = (const unsigned char*)q; p.k = (const unsigned char*)k16; p.v = (const unsigned char*)v16;
    p.out = out; p.lse = lse;
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.Sq = Sq; p.Skv = Skv;
    p.q_rs = 2L * D; p.q_hs = p.q_rs * Sq; p.q_bs = p.q_hs * Hq;
    p.o_bs = p.q_bs; p.o_hs = p.q_hs; p.o_rs = p.q_rs;
    if (strides) {
        p.q_bs = 2 * strides[0]; p.q_hs = 2 * strides[1]; p.q_rs = 2 * strides[2];
        p.o_bs = 2 * strides[3]; p.o_hs = 2 * strides[4]; p.o_rs = 2 * strides[5];
    }
    p.nqb = ceil_div(Sq, kQPerWG16);
    p.ntiles = ceil_div(Skv, 32);
    p.xcd_remap = ((B * Hq) % 8 == 0 && xcd_count() == 8) ? 1 : 0;   // (the block map is written for 8 XCDs, qattn_attn.h)
    const float sm = sm_scale > 0.0f ? sm_scale : 1.0f / sqrtf((float)D);
    p.sm_log2e = sm * 1.4426950408889634f;
    p.fast_exp = fast_exp != 0;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (D == 64) rc = fmt == QATTN_FMT_BF16 ? launch16<64, QATTN_FMT_BF16>(p, is_causal, st) : launch16<64, QATTN_FMT_FP16>(p, is_causal, st);
    else if (D == 128) rc = fmt == QATTN_FMT_BF16 ? launch16<128, QATTN_FMT_BF16>(p, is_causal, st) : launch16<128, QATTN_FMT_FP16>(p, is_causal, st);
    else rc = fmt == QATTN_FMT_BF16 ? launch16<256, QATTN_FMT_BF16>(p, is_causal, st) : launch16<256, QATTN_FMT_FP16>(p, is_causal, st);
    if (rc != QATTN_OK) return rc;
    return hipGetLastError() == hipSuccess ? QATTN_OK : QATTN_ERR_LAUNCH;
}

extern "C" int qattn_attention_forward_16(const void* q, const void* k16, const void* v16, void* out, float* lse, int B, int Hq,
                                          int Hkv, int Sq, int Skv, int D, int fmt, int is_causal, float sm_scale, int fast_exp,
                                          void* stream) {
    return qattn_attention_forward_16_strided(q, nullptr, k16, v16, out, lse, B, Hq, Hkv, Sq, Skv, D, fmt, is_causal, sm_scale, fast_exp, stream);
}
