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
//   V16FRAG chunk = [m:D/32][t:2][s:2][hh:2][d:32][8 x 16 bit] piece j = V[64c + 32t + 16s + 8(j>>2) + 4hh + (j&3)][32m + d]
// so a chunk is a linear LDS-DMA copy and every MFMA A operand is one conflict-free ds_read_b128.
// Structure: 8 waves x 32 rows, 3-stage LDS ring, one barrier per chunk (the first, non-pipelined fp8 structure);
// exact exp2, fp32 row sums, deferred rescale.
#include "qattn_attn.h"

namespace qattn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int FMT16> struct T16;
template <> struct T16<QATTN_FMT_BF16> {
    typedef __bf16 elt;
    typedef bf16x8 vec;
    static __device__ __forceinline__ v16f mfma(vec a, vec b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct T16<QATTN_FMT_FP16> {
    typedef _Float16 elt;
    typedef f16x8 vec;
    static __device__ __forceinline__ v16f mfma(vec a, vec b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

struct Attn16Params {
    const unsigned char* q;  // [B,Hq,Sq,D] 16-bit row-major
    const unsigned char* k;  // K16FRAG
    const unsigned char* v;  // V16FRAG
    void* out;
    float* lse;
    int B, Hq, Hkv, Sq, Skv;
    int nqb, nchunks, xcd_remap;
    float sm_log2e;
};

constexpr int kStages16 = 3;

template <int D>
__device__ __forceinline__ void stage16(const unsigned char* kg, const unsigned char* vg, unsigned char* lds_stage, int wave, int lane) {
    constexpr int CH = 64 * D * 2;                       // bytes of one K (or V) chunk
    constexpr int ROUNDS = 2 * CH / (kThreads * 16);
    const int wave_base = wave << 10;
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const int o = r * (kThreads * 16) + wave_base;
        const unsigned char* src = (o < CH ? kg + o : vg + (o - CH)) + (lane << 4);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds_stage + o), 16, 0, 0);
    }
}

template <int D, int FMT16, bool CAUSAL>
__global__ __launch_bounds__(kThreads, 2) void attn16_fwd_kernel(const Attn16Params p) {
    typedef typename T16<FMT16>::vec vec16;
    typedef typename T16<FMT16>::elt elt16;
    constexpr int CH = 64 * D * 2, STAGE = 2 * CH;
    constexpr int KS = D / 16;   // QK^T k-steps
    constexpr int MB = D / 32;   // O^T row blocks
    constexpr int ROUNDS = 2 * CH / (kThreads * 16);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hh = lane >> 5;
    int bid = blockIdx.x, head, qb;
    if (p.xcd_remap) {
        const int xcd = bid & 7, idx = bid >> 3;
        head = xcd * ((p.B * p.Hq) >> 3) + idx / p.nqb;
        qb = idx % p.nqb;
    } else {
        head = bid / p.nqb;
        qb = bid % p.nqb;
    }
    if (CAUSAL) qb = p.nqb - 1 - qb;
    const int b = head / p.Hq, h = head % p.Hq;
    const long kv_head = (long)b * p.Hkv + h / (p.Hq / p.Hkv);
    const int q0_wg = qb * kQPerWG, q0 = q0_wg + wave * kQPerWave, qrow = q0 + ql;
    const unsigned char* kg = p.k + kv_head * (long)p.nchunks * CH;
    const unsigned char* vg = p.v + kv_head * (long)p.nchunks * CH;
    int nloop = p.nchunks;
    if (CAUSAL) nloop = min(nloop, (min(q0_wg + kQPerWG, p.Sq) - 1) / 64 + 1);

    stage16<D>(kg, vg, smem, wave, lane);
    if (nloop > 1) stage16<D>(kg + CH, vg + CH, smem + STAGE, wave, lane);

    // Q^T fragments: lane (q, hh) holds Q[q][16s + 8hh + (0..7)] for every k-step s
    vec16 qf[KS];
    {
        const bool qvalid = qrow < p.Sq;
        const unsigned char* qp = p.q + ((((long)b * p.Hq + h) * p.Sq + (qvalid ? qrow : 0)) * D + hh * 8) * 2;
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
    float m_run = -INFINITY, l_run = 0.0f;
    const int frag_lane_off = (hh << 9) + (ql << 4);

    for (int c_idx = 0; c_idx < nloop; c_idx++) {
        if (c_idx + 1 < nloop) { if (ROUNDS == 2) wait_vmcnt<2>(); else wait_vmcnt<4>(); }
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (c_idx + 2 < nloop)
            stage16<D>(kg + (long)(c_idx + 2) * CH, vg + (long)(c_idx + 2) * CH, smem + ((c_idx + 2) % kStages16) * STAGE, wave, lane);
        const int k0 = c_idx * 64;
        if (CAUSAL && k0 > q0 + kQPerWave - 1) continue;  // fully masked for this wave
        const unsigned char* kbuf = smem + (c_idx % kStages16) * STAGE + frag_lane_off;
        const unsigned char* vbuf = kbuf + CH;

        // ---- S^T = K . Q^T (two 32-key tiles, D/16 k-steps each)
        v16f s0, s1;
#pragma unroll
        for (int r = 0; r < 16; r++) { s0[r] = 0.0f; s1[r] = 0.0f; }
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const vec16 ka = *reinterpret_cast<const vec16*>(kbuf + ((0 * KS + s) << 10));
            const vec16 kb = *reinterpret_cast<const vec16*>(kbuf + ((1 * KS + s) << 10));
            s0 = T16<FMT16>::mfma(ka, qf[s], s0);
            s1 = T16<FMT16>::mfma(kb, qf[s], s1);
        }
        float sc[32];
#pragma unroll
        for (int r = 0; r < 16; r++) { sc[r] = s0[r]; sc[16 + r] = s1[r]; }
        const bool need_mask = (k0 + 64 > p.Skv) || (CAUSAL && k0 + 63 > q0);
        if (need_mask) {
#pragma unroll
            for (int r = 0; r < 32; r++) {
                const int key = k0 + 32 * (r >> 4) + (r & 3) + 8 * ((r & 15) >> 2) + 4 * hh;
                const bool dead = key >= p.Skv || (CAUSAL && key > qrow);
                sc[r] = dead ? -INFINITY : sc[r];
            }
        }
        float mx = sc[0];
#pragma unroll
        for (int r = 1; r < 32; r++) mx = fmaxf(mx, sc[r]);
        {
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        const float m_new = fmaxf(m_run, mx);
        if (__any((m_new - m_run) * c > kRescaleThr)) {  // deferred rescale (always on the first chunk)
            const float alpha = (m_new == m_run) ? 1.0f : __builtin_amdgcn_exp2f((m_run - m_new) * c);
#pragma unroll
            for (int m = 0; m < MB; m++)
#pragma unroll
                for (int r = 0; r < 16; r++) o[m][r] *= alpha;
            l_run *= alpha;
            m_run = m_new;
        }
        const float mc = -m_run * c;
        // ---- P = exp2(c*s - c*m) in fp32, row sums in fp32, then 16-bit conversion -> PV B operands
        float pr[32];
        float ls = 0.0f;
#pragma unroll
        for (int r = 0; r < 32; r++) { pr[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], c, mc)); ls += pr[r]; }
        l_run += ls;
        vec16 pb[4];  // [tile t][k-step s'] -> index 2t + s'
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) pb[i][j] = (elt16)pr[8 * i + j];
        // ---- O^T += V^T . P^T
#pragma unroll
        for (int m = 0; m < MB; m++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const vec16 va = *reinterpret_cast<const vec16*>(vbuf + ((m * 4 + i) << 10));
                o[m] = T16<FMT16>::mfma(va, pb[i], o[m]);
            }
    }

    float l_tot;
    {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
        l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const float inv = 1.0f / l_tot;
    if (qrow < p.Sq) {
        elt16* op = reinterpret_cast<elt16*>(p.out) + (((long)b * p.Hq + h) * p.Sq + qrow) * D;
#pragma unroll
        for (int m = 0; m < MB; m++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                typedef elt16 e4 __attribute__((ext_vector_type(4)));
                e4 t;
#pragma unroll
                for (int i = 0; i < 4; i++) t[i] = (elt16)(o[m][4 * j + i] * inv);
                *reinterpret_cast<e4*>(op + 32 * m + 8 * j + 4 * hh) = t;
            }
        if (p.lse && hh == 0) p.lse[((long)b * p.Hq + h) * p.Sq + qrow] = 0.6931471805599453f * (m_run * c) + __logf(l_tot);
    }
}

// 16-bit row-major [G, S, D] -> K16FRAG / V16FRAG.  grid = (ceil(S/64), G), block = 256.
template <int D, int LAYOUT>
__global__ __launch_bounds__(256) void pack16_tile_kernel(const uint4* __restrict__ x, uint4* __restrict__ out, int S) {
    constexpr int VPR = D / 8;            // 16-byte vectors (8 elements) per row
    constexpr int RSTRIDE = D * 2 + 4;    // V staging: row stride in bytes (breaks the power-of-two stride for the gather)
    __shared__ __attribute__((aligned(16))) unsigned char img[64 * RSTRIDE];
    const int g = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x, row0 = tile * 64;
    const uint4* xg = x + (long)g * S * VPR;
    const long Sp = (long)((S + 63) / 64) * 64;
    uint4* og = out + ((long)g * Sp + row0) * VPR;   // a chunk is 64*D*2 bytes = 64*VPR vectors
    if (LAYOUT == QATTN_LAYOUT_K16FRAG) {
        // piece (t, s, hh, key) = K[32t + key][16s + 8hh .. +7] is one 16-byte vector of the source row: pure re-indexing
        for (int i = tid; i < 64 * VPR; i += 256) {
            const int key = i & 31, hh2 = (i >> 5) & 1, s = (i >> 6) % (D / 16), t = i / (64 * (D / 16));
            const int row = row0 + 32 * t + key;
            uint4 raw = make_uint4(0, 0, 0, 0);
            if (row < S) raw = xg[(long)row * VPR + 2 * s + hh2];
            og[i] = raw;
        }
    } else {
        for (int vecn = tid; vecn < 64 * VPR; vecn += 256) {
            const int r = vecn / VPR, dv = vecn % VPR, row = row0 + r;
            uint4 raw = make_uint4(0, 0, 0, 0);
            if (row < S) raw = xg[(long)row * VPR + dv];
            unsigned* dst = reinterpret_cast<unsigned*>(img + r * RSTRIDE + dv * 16);
            dst[0] = raw.x; dst[1] = raw.y; dst[2] = raw.z; dst[3] = raw.w;
        }
        __syncthreads();
        // output vector n: [m:D/32][t:2][s:2][hh:2][d:32]; its element j = V[32t + 16s + 8(j>>2) + 4hh + (j&3)][32m + d]
        for (int n = tid; n < 64 * VPR; n += 256) {
            const int dl = n & 31, hh2 = (n >> 5) & 1, s = (n >> 6) & 1, t = (n >> 7) & 1, m = n >> 8;
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

template <int D, int FMT16>
static int launch16(const Attn16Params& p, int causal, hipStream_t st) {
    const int grid = p.B * p.Hq * p.nqb;
    const size_t lds = (size_t)kStages16 * 2 * 64 * D * 2;
    if (causal) {
        auto kern = attn16_fwd_kernel<D, FMT16, true>;
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return QATTN_ERR_LAUNCH;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, st, p);
    } else {
        auto kern = attn16_fwd_kernel<D, FMT16, false>;
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return QATTN_ERR_LAUNCH;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, st, p);
    }
    return QATTN_OK;
}

}  // namespace qattn

using namespace qattn;

extern "C" size_t qattn_16bit_tensor_bytes(int layout, int B, int H, int S, int D) {
    if (B <= 0 || H <= 0 || S <= 0 || D <= 0) return 0;
    const size_t Sp = layout == QATTN_LAYOUT_ROWMAJOR ? (size_t)S : (size_t)((S + 63) / 64) * 64;
    return (size_t)B * H * Sp * D * 2;
}

extern "C" int qattn_pack16(const void* x_rowmajor, void* x_packed, int B, int H, int S, int D, int out_layout, void* stream) {
    if (!x_rowmajor || !x_packed || B <= 0 || H <= 0 || S <= 0) return QATTN_ERR_INVALID_ARG;
    if (D != 64 && D != 128) return QATTN_ERR_UNSUPPORTED_DIM;
    if (out_layout != QATTN_LAYOUT_K16FRAG && out_layout != QATTN_LAYOUT_V16FRAG) return QATTN_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((S + 63) / 64, B * H), block(256);
    const uint4* xi = (const uint4*)x_rowmajor;
    uint4* xo = (uint4*)x_packed;
#define PK16(DD, LAY) hipLaunchKernelGGL((pack16_tile_kernel<DD, LAY>), grid, block, 0, st, xi, xo, S)
    if (out_layout == QATTN_LAYOUT_K16FRAG) { if (D == 64) PK16(64, QATTN_LAYOUT_K16FRAG); else PK16(128, QATTN_LAYOUT_K16FRAG); }
    else { if (D == 64) PK16(64, QATTN_LAYOUT_V16FRAG); else PK16(128, QATTN_LAYOUT_V16FRAG); }
#undef PK16
    return hipGetLastError() == hipSuccess ? QATTN_OK : QATTN_ERR_LAUNCH;
}

extern "C" int qattn_attention_forward_16(const void* q, const void* k16, const void* v16, void* out, float* lse, int B, int Hq,
                                          int Hkv, int Sq, int Skv, int D, int fmt, int is_causal, float sm_scale, void* stream) {
    if (!q || !k16 || !v16 || !out) return QATTN_ERR_INVALID_ARG;
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || Sq <= 0 || Skv <= 0) return QATTN_ERR_INVALID_ARG;
    if (D != 64 && D != 128) return QATTN_ERR_UNSUPPORTED_DIM;
    if (Hq % Hkv != 0) return QATTN_ERR_UNSUPPORTED_DIM;
    if (fmt != QATTN_FMT_BF16 && fmt != QATTN_FMT_FP16) return QATTN_ERR_UNSUPPORTED_FMT;
    Attn16Params p;
    p.q = (const unsigned char*)q; p.k = (const unsigned char*)k16; p.v = (const unsigned char*)v16;
    p.out = out; p.lse = lse;
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.Sq = Sq; p.Skv = Skv;
    p.nqb = ceil_div(Sq, kQPerWG);
    p.nchunks = ceil_div(Skv, 64);
    p.xcd_remap = ((B * Hq) % 8 == 0) ? 1 : 0;
    const float sm = sm_scale > 0.0f ? sm_scale : 1.0f / sqrtf((float)D);
    p.sm_log2e = sm * 1.4426950408889634f;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (D == 64) rc = fmt == QATTN_FMT_BF16 ? launch16<64, QATTN_FMT_BF16>(p, is_causal, st) : launch16<64, QATTN_FMT_FP16>(p, is_causal, st);
    else rc = fmt == QATTN_FMT_BF16 ? launch16<128, QATTN_FMT_BF16>(p, is_causal, st) : launch16<128, QATTN_FMT_FP16>(p, is_causal, st);
    if (rc != QATTN_OK) return rc;
    return hipGetLastError() == hipSuccess ? QATTN_OK : QATTN_ERR_LAUNCH;
}
