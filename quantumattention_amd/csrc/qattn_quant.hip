// qattn_quant.hip -- bf16/fp16 -> fp8 dynamic quantisation pre-pass and fp8 re-layout, gfx950.
//
// Replaces src/quantum_attn/nn.py:14-19 (`_dynamically_quantize_fp8`), which the reference leaves to
// Inductor-generated Triton (nn.py:22-42, 410-418).  HBM-bound byte work: 16-byte coalesced loads, the fragment
// permutation done in LDS, 16-byte linear stores.  Bit-exact to the reference in both of its numerics.
#include "qattn_common.h"

namespace qattn {

// ---------------------------------------------------------------------------------------------------------
// pass 1 (head-wise only): per-(b,h) abs-max.  |x| of bf16/fp16 is monotone in its low 15 bits, so the reduction
// runs on packed 16-bit integers; NaN payloads (> inf as integers) propagate like torch's amax.
// grid = (splits, groups), block = 256.  amax_bits[g] must be zero on entry (hipMemsetAsync in the launcher).
// ---------------------------------------------------------------------------------------------------------
template <int IN_FMT>
__global__ __launch_bounds__(256) void amax_kernel(const uint4* __restrict__ x, unsigned* __restrict__ amax_bits,
                                                   long vecs_per_group, int splits) {
    const long g = blockIdx.y;
    const uint4* xg = x + g * vecs_per_group;
    const long per = (vecs_per_group + splits - 1) / splits;
    const long beg = (long)blockIdx.x * per;
    long end = beg + per;
    if (end > vecs_per_group) end = vecs_per_group;
    unsigned m0 = 0, m1 = 0;  // packed 2x u16 running max
    for (long i = beg + threadIdx.x; i < end; i += 256) {
        uint4 v = xg[i];
        unsigned a = v.x & 0x7fff7fffu, b = v.y & 0x7fff7fffu, c = v.z & 0x7fff7fffu, d = v.w & 0x7fff7fffu;
        // packed u16 max
        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
        u16x2 pa, pb, pc, pd, p0, p1;
        __builtin_memcpy(&pa, &a, 4); __builtin_memcpy(&pb, &b, 4); __builtin_memcpy(&pc, &c, 4); __builtin_memcpy(&pd, &d, 4);
        __builtin_memcpy(&p0, &m0, 4); __builtin_memcpy(&p1, &m1, 4);
        p0 = __builtin_elementwise_max(p0, __builtin_elementwise_max(pa, pb));
        p1 = __builtin_elementwise_max(p1, __builtin_elementwise_max(pc, pd));
        __builtin_memcpy(&m0, &p0, 4); __builtin_memcpy(&m1, &p1, 4);
    }
    unsigned m = max(max(m0 & 0xffffu, m0 >> 16), max(m1 & 0xffffu, m1 >> 16));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    __shared__ unsigned red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(red[0], red[1]), max(red[2], red[3]));
        const float f = load16f<IN_FMT>((unsigned short)m);  // |amax| as fp32 (exact)
        atomicMax(amax_bits + g, __float_as_uint(f));         // non-negative floats order like their bit patterns
    }
}


// Transposing copy-out of one staged fp8 tile (64 keys, row-major with VSTRIDE-byte rows in LDS) into its VFRAG chunk.
// A thread owns 8 keys (the two 4-key groups w = 2wh, 2wh+1 of one (half, hh)) x 4 consecutive d: 8 conflict-free
// ds_read_b32, two 4x4 byte transposes (8 v_perm_b32 each) and four 8-byte stores -- instead of 32 ds_read_u8 and
// 24 shift/or per 8 output dwords.
template <int D, int VSTRIDE>
__device__ __forceinline__ void vfrag_copy_out(const unsigned char* img, unsigned char* og_chunk, int tid) {
#pragma unroll
    for (int k = 0; k < (2 * D + 255) / 256; k++) {
        const int blk = k * 256 + tid;
        if (2 * D < 256 && blk >= 2 * D) break;
        // lane bits chosen so that the 32 lanes of a ds_read_b32 group hit 32 different banks of the 33-dword-stride image:
        // bank = (33*row + 8m + dq) mod 32 = dq + 4hh + 16wh + 8(m&1) + const over (dq&3, hh, wh, m&1)
        const int wh = blk & 1, hh = (blk >> 3) & 1, half = (blk >> 6) & 1;
        const int dq = ((blk >> 1) & 3) + 4 * ((blk >> 5) & 1), m = ((blk >> 4) & 1) + 2 * (blk >> 7);
        const int d0 = 32 * m + 4 * dq;
        unsigned o[2][4];
#pragma unroll
        for (int wi = 0; wi < 2; wi++) {
            const unsigned char* src = img + (32 * half + 8 * (2 * wh + wi) + 4 * hh) * VSTRIDE + d0;
            const unsigned r0 = *reinterpret_cast<const unsigned*>(src), r1 = *reinterpret_cast<const unsigned*>(src + VSTRIDE);
            const unsigned r2 = *reinterpret_cast<const unsigned*>(src + 2 * VSTRIDE), r3 = *reinterpret_cast<const unsigned*>(src + 3 * VSTRIDE);
            const unsigned t0 = __builtin_amdgcn_perm(r1, r0, 0x05010400u), t1 = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
            const unsigned t2 = __builtin_amdgcn_perm(r3, r2, 0x05010400u), t3 = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
            o[wi][0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
            o[wi][1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
            o[wi][2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
            o[wi][3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
        }
        unsigned char* dst = og_chunk + ((((m * 2 + hh) * 2 + half) * 32 + 4 * dq) << 4) + 8 * wh;
#pragma unroll
        for (int j = 0; j < 4; j++) *reinterpret_cast<uint2*>(dst + 16 * j) = make_uint2(o[0][j], o[1][j]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// pass 2: quantise one tile of 64 rows x D and emit it in the requested layout.
// grid = (ceil(S/64), B*H), block = 256.  Each thread owns D/32 vectors of 8 consecutive elements of one row.
// ---------------------------------------------------------------------------------------------------------
template <int D, int IN_FMT, int OUT_FMT, int LAYOUT, bool TOKEN>
__global__ __launch_bounds__(256) void quant_tile_kernel(const uint4* __restrict__ x, uint4* __restrict__ out,
                                                         float* __restrict__ scale_out,
                                                         const unsigned* __restrict__ amax_bits, int S, int numerics) {
    constexpr int VPR = D / 8;            // 16-byte input vectors per row
    constexpr int ITERS = 64 * VPR / 256; // vectors per thread
    constexpr int VSTRIDE = D + 4;  // VFRAG staging: fp8 row-major with 4 bytes of row padding (bank spread)
    __shared__ __attribute__((aligned(16))) unsigned char img[LAYOUT == QATTN_LAYOUT_VFRAG ? 64 * VSTRIDE : 64 * D];
    const int g = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const int row0 = tile * 64;
    const float qmax = OUT_FMT == QATTN_FMT_E4M3 ? 448.0f : 57344.0f;
    const float inv_qmax = (float)(1.0 / (double)(OUT_FMT == QATTN_FMT_E4M3 ? 448.0 : 57344.0));
    float scale = 1.0f;
    if (!TOKEN) {
        scale = make_scale(__uint_as_float(amax_bits[g]), inv_qmax, numerics, IN_FMT);
        if (tile == 0 && tid == 0) scale_out[g] = scale;
    }
    float rinv = 1.0f / scale;
    (void)qmax;
    const uint4* xg = x + (long)g * S * VPR;
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int vec = it * 256 + tid;
        const int r = vec / VPR, dv = vec % VPR;
        const int row = row0 + r;
        uint4 raw = make_uint4(0, 0, 0, 0);
        if (row < S) raw = xg[(long)row * VPR + dv];
        unsigned short e[8];
        __builtin_memcpy(e, &raw, 16);
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; j++) f[j] = load16f<IN_FMT>(e[j]);
        if (TOKEN) {
            float a = 0.0f;
            bool nan = false;
#pragma unroll
            for (int j = 0; j < 8; j++) { a = fmaxf(a, fabsf(f[j])); nan |= (f[j] != f[j]); }
            unsigned ab = nan ? 0x7fc00000u : __float_as_uint(a);
#pragma unroll
            for (int off = VPR / 2; off > 0; off >>= 1) ab = max(ab, (unsigned)__shfl_xor((int)ab, off));
            scale = make_scale(__uint_as_float(ab), inv_qmax, numerics, IN_FMT);
            if (dv == 0 && row < S) scale_out[(long)g * S + row] = scale;
        }
        if (TOKEN) rinv = 1.0f / scale;
        const int2 lohi = quant8<IN_FMT, OUT_FMT>(raw, scale, rinv);
        const int lo = lohi.x, hi = lohi.y;
        const int d0 = dv * 8;
        if (LAYOUT == QATTN_LAYOUT_ROWMAJOR) {
            *reinterpret_cast<int2*>(img + r * D + d0) = make_int2(lo, hi);
        } else if (LAYOUT == QATTN_LAYOUT_KFRAG) {
            *reinterpret_cast<int2*>(img + kfrag_offset<D>(r, d0)) = make_int2(lo, hi);
        } else {
            *reinterpret_cast<int*>(img + r * VSTRIDE + d0) = lo;
            *reinterpret_cast<int*>(img + r * VSTRIDE + d0 + 4) = hi;
        }
    }
    __syncthreads();
    // linear write-out of the 64*D-byte image
    constexpr int OUT_VECS = 64 * D / 16;
    if (LAYOUT == QATTN_LAYOUT_ROWMAJOR) {
        uint4* og = out + (long)g * S * (D / 16) + (long)row0 * (D / 16);
        const int valid = (S - row0 < 64 ? S - row0 : 64) * (D / 16);
        for (int i = tid; i < valid; i += 256) og[i] = reinterpret_cast<const uint4*>(img)[i];
    } else if (LAYOUT == QATTN_LAYOUT_KFRAG) {
        const long Sp = (long)((S + 63) / 64) * 64;
        uint4* og = out + ((long)g * Sp + row0) * (D / 16);
        for (int i = tid; i < OUT_VECS; i += 256) og[i] = reinterpret_cast<const uint4*>(img)[i];
    } else {
        const long Sp = (long)((S + 63) / 64) * 64;
        vfrag_copy_out<D, VSTRIDE>(img, reinterpret_cast<unsigned char*>(out) + ((long)g * Sp + row0) * D, tid);
    }
}

// fp8 row-major -> fragment layout (pure byte permutation).  grid = (ceil(S/64), B*H), block = 256.
template <int D, int LAYOUT>
__global__ __launch_bounds__(256) void pack_tile_kernel(const uint4* __restrict__ x, uint4* __restrict__ out, int S) {
    constexpr int VPR = D / 16;
    __shared__ __attribute__((aligned(16))) unsigned char img[64 * D];
    const int g = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x, row0 = tile * 64;
    const uint4* xg = x + (long)g * S * VPR;
    for (int vec = tid; vec < 64 * VPR; vec += 256) {
        const int r = vec / VPR, dv = vec % VPR, row = row0 + r;
        uint4 raw = make_uint4(0, 0, 0, 0);
        if (row < S) raw = xg[(long)row * VPR + dv];
        if (LAYOUT == QATTN_LAYOUT_KFRAG) {
            *reinterpret_cast<uint4*>(img + kfrag_offset<D>(r, dv * 16)) = raw;
        } else {
            unsigned char b[16];
            __builtin_memcpy(b, &raw, 16);
#pragma unroll
            for (int j = 0; j < 16; j++) img[vfrag_offset<D>(r, dv * 16 + j)] = b[j];
        }
    }
    __syncthreads();
    const long Sp = (long)((S + 63) / 64) * 64;
    uint4* og = out + ((long)g * Sp + row0) * (D / 16);
    for (int i = tid; i < 64 * D / 16; i += 256) og[i] = reinterpret_cast<const uint4*>(img)[i];
}

// ---------------------------------------------------------------------------------------------------------
// Fused q/k/v pre-pass: ONE amax launch and ONE quantise launch cover the three tensors (blockIdx.z = tensor).
// LDS patterns are conflict-free (the first version's byte scatter spent ~90 % of its LDS cycles in bank
// conflicts, profiles/r01_v1/pmc_summary.json):
//   ROWMAJOR  no LDS at all: 8-byte coalesced stores.
//   KFRAG     8-byte writes into the fragment image padded by 16 B per 512 B, linear 16-byte copy-out.
//   VFRAG     fp8 tile row-major in LDS with a 132-byte row stride, then each thread gathers the 4 keys of an
//             output dword with ds_read_u8 (4 lanes share a dword = broadcast, 16 banks hit per instruction) and
//             stores dwords straight to global memory.
// ---------------------------------------------------------------------------------------------------------
struct QuantJob {
    const uint4* x;       // [G, S, D] 16-bit
    uint4* out;           // fp8 payload
    float* scale;         // [G] (head) or [G, S] (token)
    unsigned* amax_bits;  // [G] workspace (head-wise)
    int G, S, layout, token;
};
struct QuantJobs {
    QuantJob j[3];
};

template <int IN_FMT>
__global__ __launch_bounds__(256) void amax_multi_kernel(const QuantJobs jobs, int D, int splits, int zbase) {
    const QuantJob& jb = jobs.j[zbase + blockIdx.z];
    if (jb.token || (int)blockIdx.y >= jb.G) return;
    const long vecs_per_group = (long)jb.S * D / 8;
    const long g = blockIdx.y;
    const uint4* xg = jb.x + g * vecs_per_group;
    const long per = (vecs_per_group + splits - 1) / splits;
    const long beg = (long)blockIdx.x * per;
    long end = beg + per;
    if (end > vecs_per_group) end = vecs_per_group;
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    unsigned m0 = 0, m1 = 0;
    auto fold = [&](const uint4& v) {
        unsigned a = v.x & 0x7fff7fffu, b = v.y & 0x7fff7fffu, c = v.z & 0x7fff7fffu, d = v.w & 0x7fff7fffu;
        u16x2 pa, pb, pc, pd, p0, p1;
        __builtin_memcpy(&pa, &a, 4); __builtin_memcpy(&pb, &b, 4); __builtin_memcpy(&pc, &c, 4); __builtin_memcpy(&pd, &d, 4);
        __builtin_memcpy(&p0, &m0, 4); __builtin_memcpy(&p1, &m1, 4);
        p0 = __builtin_elementwise_max(p0, __builtin_elementwise_max(pa, pb));
        p1 = __builtin_elementwise_max(p1, __builtin_elementwise_max(pc, pd));
        __builtin_memcpy(&m0, &p0, 4); __builtin_memcpy(&m1, &p1, 4);
    };
    // 8 independent 16-byte loads in flight per thread (a plain strided loop kept ~2 and ran at 4.8 TB/s)
    long i = beg + threadIdx.x;
    for (; i + 7 * 256 < end; i += 8 * 256) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = xg[i + u * 256];
#pragma unroll
        for (int u = 0; u < 8; u++) fold(v[u]);
    }
    for (; i < end; i += 256) fold(xg[i]);
    unsigned m = max(max(m0 & 0xffffu, m0 >> 16), max(m1 & 0xffffu, m1 >> 16));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    __shared__ unsigned red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(red[0], red[1]), max(red[2], red[3]));
        atomicMax(jb.amax_bits + g, __float_as_uint(load16f<IN_FMT>((unsigned short)m)));
    }
}

template <int D, int IN_FMT, int OUT_FMT>
__global__ __launch_bounds__(256) void quant_multi_kernel(const QuantJobs jobs, int numerics, int ztop) {
    constexpr int VPR = D / 8;             // 16-byte input vectors per row
    constexpr int ITERS = 64 * VPR / 256;  // vectors per thread
    constexpr int KPAD = 64 * D + (64 * D / 512) * 16;  // KFRAG image + 16 B per 512 B
    constexpr int VSTRIDE = D + 4;                      // VFRAG staging: fp8 row-major, 132-byte rows for D = 128
    constexpr int LDS_BYTES = KPAD > 64 * VSTRIDE ? KPAD : 64 * VSTRIDE;
    __shared__ __attribute__((aligned(16))) unsigned char img[LDS_BYTES];
    // walk the tensors, heads and tiles in the REVERSE order of the amax pass: the amax pass streamed 3 tensors
    // through the 256 MiB Infinity Cache, so its last ~256 MiB are the bytes most likely still on-die
    const QuantJob& jb = jobs.j[ztop - blockIdx.z];
    const int tid = threadIdx.x;
    const int S = jb.S;
    const int g = jb.G - 1 - (int)blockIdx.y, tile = (S + 63) / 64 - 1 - (int)blockIdx.x;
    const int row0 = tile * 64;
    if (g < 0 || tile < 0) return;
    const int layout = jb.layout;
    const bool token = jb.token != 0;
    const float qmax = OUT_FMT == QATTN_FMT_E4M3 ? 448.0f : 57344.0f;
    const float inv_qmax = (float)(1.0 / (double)(OUT_FMT == QATTN_FMT_E4M3 ? 448.0 : 57344.0));
    float scale = 1.0f;
    if (!token) {
        scale = make_scale(__uint_as_float(jb.amax_bits[g]), inv_qmax, numerics, IN_FMT);
        if (tile == 0 && tid == 0) jb.scale[g] = scale;
    }
    float rinv = 1.0f / scale;
    (void)qmax;
    const uint4* xg = jb.x + (long)g * S * VPR;
    const long Sp = (long)((S + 63) / 64) * 64;
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int vec = it * 256 + tid;
        const int r = vec / VPR, dv = vec % VPR;
        const int row = row0 + r;
        uint4 raw = make_uint4(0, 0, 0, 0);
        if (row < S) raw = xg[(long)row * VPR + dv];
        unsigned short e[8];
        __builtin_memcpy(e, &raw, 16);
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; j++) f[j] = load16f<IN_FMT>(e[j]);
        if (token) {
            float a = 0.0f;
            bool nan = false;
#pragma unroll
            for (int j = 0; j < 8; j++) { a = fmaxf(a, fabsf(f[j])); nan |= (f[j] != f[j]); }
            unsigned ab = nan ? 0x7fc00000u : __float_as_uint(a);
#pragma unroll
            for (int off = VPR / 2; off > 0; off >>= 1) ab = max(ab, (unsigned)__shfl_xor((int)ab, off));
            scale = make_scale(__uint_as_float(ab), inv_qmax, numerics, IN_FMT);
            if (dv == 0 && row < S) jb.scale[(long)g * S + row] = scale;
        }
        if (token) rinv = 1.0f / scale;
        const int2 lohi = quant8<IN_FMT, OUT_FMT>(raw, scale, rinv);
        const int lo = lohi.x, hi = lohi.y;
        const int d0 = dv * 8;
        if (layout == QATTN_LAYOUT_ROWMAJOR) {
            if (row < S) reinterpret_cast<int2*>(jb.out)[((long)g * S + row) * (D / 8) + dv] = make_int2(lo, hi);
        } else if (layout == QATTN_LAYOUT_KFRAG) {
            const int o = kfrag_offset<D>(r, d0);
            *reinterpret_cast<int2*>(img + o + ((o >> 9) << 4)) = make_int2(lo, hi);
        } else {
            *reinterpret_cast<int*>(img + r * VSTRIDE + d0) = lo;
            *reinterpret_cast<int*>(img + r * VSTRIDE + d0 + 4) = hi;
        }
    }
    if (layout == QATTN_LAYOUT_ROWMAJOR) return;
    __syncthreads();
    if (layout == QATTN_LAYOUT_KFRAG) {
        uint4* og = jb.out + ((long)g * Sp + row0) * (D / 16);
        for (int i = tid; i < 64 * D / 16; i += 256) og[i] = *reinterpret_cast<const uint4*>(img + i * 16 + ((i >> 5) << 4));
    } else {
        vfrag_copy_out<D, VSTRIDE>(img, reinterpret_cast<unsigned char*>(jb.out) + ((long)g * Sp + row0) * D, tid);
    }
}

// ---------------------------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------------------------
template <int D, int IN_FMT, int OUT_FMT>
static int launch_quant_dl(const void* x, void* x8, float* scale, int G, int S, int scale_mode, int numerics,
                           int layout, const unsigned* amax, hipStream_t st) {
    dim3 grid((S + 63) / 64, G), block(256);
    const uint4* xi = (const uint4*)x;
    uint4* xo = (uint4*)x8;
#define QL(LAY, TOK) hipLaunchKernelGGL((quant_tile_kernel<D, IN_FMT, OUT_FMT, LAY, TOK>), grid, block, 0, st, xi, xo, scale, amax, S, numerics)
    const bool tok = scale_mode == QATTN_SCALE_TOKEN;
    if (layout == QATTN_LAYOUT_ROWMAJOR) { if (tok) QL(QATTN_LAYOUT_ROWMAJOR, true); else QL(QATTN_LAYOUT_ROWMAJOR, false); }
    else if (layout == QATTN_LAYOUT_KFRAG) { if (tok) QL(QATTN_LAYOUT_KFRAG, true); else QL(QATTN_LAYOUT_KFRAG, false); }
    else if (layout == QATTN_LAYOUT_VFRAG) { if (tok) QL(QATTN_LAYOUT_VFRAG, true); else QL(QATTN_LAYOUT_VFRAG, false); }
    else return QATTN_ERR_INVALID_ARG;
#undef QL
    return QATTN_OK;
}

template <int D>
static int launch_quant_d(const void* x, int in_fmt, void* x8, float* scale, int G, int S, int out_fmt, int scale_mode,
                          int numerics, int layout, const unsigned* amax, hipStream_t st) {
    if (in_fmt == QATTN_FMT_BF16 && out_fmt == QATTN_FMT_E4M3) return launch_quant_dl<D, QATTN_FMT_BF16, QATTN_FMT_E4M3>(x, x8, scale, G, S, scale_mode, numerics, layout, amax, st);
    if (in_fmt == QATTN_FMT_BF16 && out_fmt == QATTN_FMT_E5M2) return launch_quant_dl<D, QATTN_FMT_BF16, QATTN_FMT_E5M2>(x, x8, scale, G, S, scale_mode, numerics, layout, amax, st);
    if (in_fmt == QATTN_FMT_FP16 && out_fmt == QATTN_FMT_E4M3) return launch_quant_dl<D, QATTN_FMT_FP16, QATTN_FMT_E4M3>(x, x8, scale, G, S, scale_mode, numerics, layout, amax, st);
    if (in_fmt == QATTN_FMT_FP16 && out_fmt == QATTN_FMT_E5M2) return launch_quant_dl<D, QATTN_FMT_FP16, QATTN_FMT_E5M2>(x, x8, scale, G, S, scale_mode, numerics, layout, amax, st);
    return QATTN_ERR_UNSUPPORTED_FMT;
}

}  // namespace qattn

using namespace qattn;

extern "C" size_t qattn_fp8_tensor_bytes(int layout, int B, int H, int S, int D) {
    if (B <= 0 || H <= 0 || S <= 0 || D <= 0) return 0;
    const size_t Sp = layout == QATTN_LAYOUT_ROWMAJOR ? (size_t)S : (size_t)((S + 63) / 64) * 64;
    return (size_t)B * H * Sp * D;
}

extern "C" size_t qattn_quant_workspace_bytes(int B, int H, int S, int D, int scale_mode) {
    (void)S; (void)D;
    if (B <= 0 || H <= 0) return 0;
    return scale_mode == QATTN_SCALE_HEAD ? (size_t)B * H * sizeof(unsigned) : 0;
}

extern "C" int qattn_quant_fp8(const void* x, int in_fmt, void* x8, float* scale, int B, int H, int S, int D,
                               int out_fmt, int scale_mode, int numerics, int out_layout, void* workspace,
                               size_t workspace_bytes, void* stream) {
    if (!x || !x8 || !scale || B <= 0 || H <= 0 || S <= 0) return QATTN_ERR_INVALID_ARG;
    if (D != 64 && D != 128 && D != 256) return QATTN_ERR_UNSUPPORTED_DIM;
    if (scale_mode != QATTN_SCALE_HEAD && scale_mode != QATTN_SCALE_TOKEN) return QATTN_ERR_INVALID_ARG;
    if (numerics != QATTN_NUMERICS_COMPILED && numerics != QATTN_NUMERICS_EAGER) return QATTN_ERR_INVALID_ARG;
    if (in_fmt != QATTN_FMT_BF16 && in_fmt != QATTN_FMT_FP16) return QATTN_ERR_UNSUPPORTED_FMT;
    if (out_fmt != QATTN_FMT_E4M3 && out_fmt != QATTN_FMT_E5M2) return QATTN_ERR_UNSUPPORTED_FMT;
    hipStream_t st = (hipStream_t)stream;
    const int G = B * H;
    unsigned* amax = nullptr;
    if (scale_mode == QATTN_SCALE_HEAD && (!workspace || workspace_bytes < (size_t)G * sizeof(unsigned))) return QATTN_ERR_WORKSPACE;
    if (scale_mode == QATTN_SCALE_HEAD) {
        amax = (unsigned*)workspace;
        if (hipMemsetAsync(amax, 0, (size_t)G * sizeof(unsigned), st) != hipSuccess) return QATTN_ERR_LAUNCH;
        const long vecs = (long)S * D / 8;
        int splits = (int)((vecs + 2047) / 2048);  // 8 x 16 B per thread and block  // >= 16 vectors per thread per split
        if (splits < 1) splits = 1;
        if (splits > 256) splits = 256;
        dim3 grid(splits, G), block(256);
        if (in_fmt == QATTN_FMT_BF16) hipLaunchKernelGGL((amax_kernel<QATTN_FMT_BF16>), grid, block, 0, st, (const uint4*)x, amax, vecs, splits);
        else hipLaunchKernelGGL((amax_kernel<QATTN_FMT_FP16>), grid, block, 0, st, (const uint4*)x, amax, vecs, splits);
    }
    int rc;
    if (D == 64) rc = launch_quant_d<64>(x, in_fmt, x8, scale, G, S, out_fmt, scale_mode, numerics, out_layout, amax, st);
    else if (D == 128) rc = launch_quant_d<128>(x, in_fmt, x8, scale, G, S, out_fmt, scale_mode, numerics, out_layout, amax, st);
    else rc = launch_quant_d<256>(x, in_fmt, x8, scale, G, S, out_fmt, scale_mode, numerics, out_layout, amax, st);
    if (rc != QATTN_OK) return rc;
    return hipGetLastError() == hipSuccess ? QATTN_OK : QATTN_ERR_LAUNCH;
}

extern "C" int qattn_pack_fp8(const void* x8_rowmajor, void* x8_packed, int B, int H, int S, int D, int out_layout,
                              void* stream) {
    if (!x8_rowmajor || !x8_packed || B <= 0 || H <= 0 || S <= 0) return QATTN_ERR_INVALID_ARG;
    if (D != 64 && D != 128 && D != 256) return QATTN_ERR_UNSUPPORTED_DIM;
    if (out_layout != QATTN_LAYOUT_KFRAG && out_layout != QATTN_LAYOUT_VFRAG) return QATTN_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((S + 63) / 64, B * H), block(256);
    const uint4* xi = (const uint4*)x8_rowmajor;
    uint4* xo = (uint4*)x8_packed;
#define PK(DD, LAY) hipLaunchKernelGGL((pack_tile_kernel<DD, LAY>), grid, block, 0, st, xi, xo, S)
    if (out_layout == QATTN_LAYOUT_KFRAG) { if (D == 64) PK(64, QATTN_LAYOUT_KFRAG); else if (D == 128) PK(128, QATTN_LAYOUT_KFRAG); else PK(256, QATTN_LAYOUT_KFRAG); }
    else { if (D == 64) PK(64, QATTN_LAYOUT_VFRAG); else if (D == 128) PK(128, QATTN_LAYOUT_VFRAG); else PK(256, QATTN_LAYOUT_VFRAG); }
#undef PK
    return hipGetLastError() == hipSuccess ? QATTN_OK : QATTN_ERR_LAUNCH;
}

extern "C" size_t qattn_quant_qkv_workspace_bytes(int B, int Hq, int Hkv) {
    if (B <= 0 || Hq <= 0 || Hkv <= 0) return 0;
    return (size_t)B * (Hq + 2 * (size_t)Hkv) * sizeof(unsigned);
}

template <int D>
static int launch_quant_multi(const QuantJobs& jobs, int in_fmt, int out_fmt, int numerics, dim3 grid, int ztop, hipStream_t st) {
    dim3 block(256);
    if (in_fmt == QATTN_FMT_BF16 && out_fmt == QATTN_FMT_E4M3) hipLaunchKernelGGL((quant_multi_kernel<D, QATTN_FMT_BF16, QATTN_FMT_E4M3>), grid, block, 0, st, jobs, numerics, ztop);
    else if (in_fmt == QATTN_FMT_BF16 && out_fmt == QATTN_FMT_E5M2) hipLaunchKernelGGL((quant_multi_kernel<D, QATTN_FMT_BF16, QATTN_FMT_E5M2>), grid, block, 0, st, jobs, numerics, ztop);
    else if (in_fmt == QATTN_FMT_FP16 && out_fmt == QATTN_FMT_E4M3) hipLaunchKernelGGL((quant_multi_kernel<D, QATTN_FMT_FP16, QATTN_FMT_E4M3>), grid, block, 0, st, jobs, numerics, ztop);
    else if (in_fmt == QATTN_FMT_FP16 && out_fmt == QATTN_FMT_E5M2) hipLaunchKernelGGL((quant_multi_kernel<D, QATTN_FMT_FP16, QATTN_FMT_E5M2>), grid, block, 0, st, jobs, numerics, ztop);
    else return QATTN_ERR_UNSUPPORTED_FMT;
    return QATTN_OK;
}

extern "C" int qattn_quant_qkv_fp8(const void* q, const void* k, const void* v, int in_fmt, void* q8, void* k8, void* v8,
                                   float* scale_q, float* scale_k, float* scale_v, int B, int Hq, int Hkv, int Sq, int Skv,
                                   int D, int out_fmt, int scale_mode, int numerics, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    if (!q || !k || !v || !q8 || !k8 || !v8 || !scale_q || !scale_k || !scale_v) return QATTN_ERR_INVALID_ARG;
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || Sq <= 0 || Skv <= 0) return QATTN_ERR_INVALID_ARG;
    if (D != 64 && D != 128 && D != 256) return QATTN_ERR_UNSUPPORTED_DIM;
    if (scale_mode != QATTN_SCALE_HEAD && scale_mode != QATTN_SCALE_TOKEN) return QATTN_ERR_INVALID_ARG;
    if (numerics != QATTN_NUMERICS_COMPILED && numerics != QATTN_NUMERICS_EAGER) return QATTN_ERR_INVALID_ARG;
    if (in_fmt != QATTN_FMT_BF16 && in_fmt != QATTN_FMT_FP16) return QATTN_ERR_UNSUPPORTED_FMT;
    if (out_fmt != QATTN_FMT_E4M3 && out_fmt != QATTN_FMT_E5M2) return QATTN_ERR_UNSUPPORTED_FMT;
    const size_t need = qattn_quant_qkv_workspace_bytes(B, Hq, Hkv);
    if (!workspace || workspace_bytes < need) return QATTN_ERR_WORKSPACE;
    return qattn::launch_quant_qkv(q, k, v, in_fmt, q8, k8, v8, scale_q, scale_k, scale_v, B, Hq, Hkv, Sq, Skv, D, out_fmt, scale_mode,
                                   numerics, (unsigned*)workspace, false, (hipStream_t)stream);
}

int qattn::launch_quant_qkv(const void* q, const void* k, const void* v, int in_fmt, void* q8, void* k8, void* v8, float* scale_q,
                            float* scale_k, float* scale_v, int B, int Hq, int Hkv, int Sq, int Skv, int D, int out_fmt,
                            int scale_mode, int numerics, unsigned* ws, bool skip_q_payload, hipStream_t st) {
    const size_t need = qattn_quant_qkv_workspace_bytes(B, Hq, Hkv);
    if (hipMemsetAsync(ws, 0, need, st) != hipSuccess) return QATTN_ERR_LAUNCH;
    const int tok = scale_mode == QATTN_SCALE_TOKEN;
    QuantJobs jobs;
    jobs.j[0] = QuantJob{(const uint4*)q, (uint4*)q8, scale_q, ws, B * Hq, Sq, QATTN_LAYOUT_ROWMAJOR, tok};
    jobs.j[1] = QuantJob{(const uint4*)k, (uint4*)k8, scale_k, ws + (size_t)B * Hq, B * Hkv, Skv, QATTN_LAYOUT_KFRAG, tok};
    jobs.j[2] = QuantJob{(const uint4*)v, (uint4*)v8, scale_v, ws + (size_t)B * (Hq + Hkv), B * Hkv, Skv, QATTN_LAYOUT_VFRAG, 0};
    // (Tried and dropped: one tensor at a time -- amax then quantise, hoping the re-read hits the 256 MiB Infinity Cache --
    // was 13 % slower than the two fused launches; a one-pass register-resident variant with a cross-workgroup amax
    // exchange was 2-6x slower, the agent-scope atomics + spinning cost more than the second read.  A third variant --
    // all slices of a head pinned to ONE XCD, abs-max exchanged through that XCD's L2, slice re-read instead of held in
    // registers -- measured 0.42 ms vs 0.18: the exchange alone costs 0.2 ms and the re-read is NOT served on-die even with
    // residency capped to 2 workgroups per CU (amax only 0.071 ms, amax + re-read without any exchange 0.19 ms).)
    const int Gmax = B * (Hq > Hkv ? Hq : Hkv), Smax = Sq > Skv ? Sq : Skv;
    {
        const long vecs = (long)Smax * D / 8;
        int splits = (int)((vecs + 2047) / 2048);  // 8 x 16 B per thread and block
        if (splits < 1) splits = 1;
        if (splits > 256) splits = 256;
        dim3 grid(splits, Gmax, 3), block(256);
        if (in_fmt == QATTN_FMT_BF16) hipLaunchKernelGGL((amax_multi_kernel<QATTN_FMT_BF16>), grid, block, 0, st, jobs, D, splits, 0);
        else hipLaunchKernelGGL((amax_multi_kernel<QATTN_FMT_FP16>), grid, block, 0, st, jobs, D, splits, 0);
    }
    // the quantise pass walks jobs ztop, ztop-1, ...: with skip_q_payload only v and k (blockIdx.z = 0, 1)
    dim3 grid(((skip_q_payload ? Skv : Smax) + 63) / 64, skip_q_payload ? B * Hkv : Gmax, skip_q_payload ? 2 : 3);
    int rc;
    if (D == 64) rc = launch_quant_multi<64>(jobs, in_fmt, out_fmt, numerics, grid, 2, st);
    else if (D == 128) rc = launch_quant_multi<128>(jobs, in_fmt, out_fmt, numerics, grid, 2, st);
    else rc = launch_quant_multi<256>(jobs, in_fmt, out_fmt, numerics, grid, 2, st);
    if (rc != QATTN_OK) return rc;
    return hipGetLastError() == hipSuccess ? QATTN_OK : QATTN_ERR_LAUNCH;
}
