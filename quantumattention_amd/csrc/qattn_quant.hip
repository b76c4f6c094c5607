// qattn_quant.hip -- bf16/fp16 -> fp8 dynamic quantisation pre-pass and fp8 re-layout, gfx950.
//
// Replaces src/quantum_attn/nn.py:14-19 (`_dynamically_quantize_fp8`), which the reference leaves to
// Inductor-generated Triton (nn.py:22-42, 410-418).  HBM-bound byte work: 16-byte coalesced loads, the fragment
// permutation done in LDS, 16-byte linear stores.  Bit-exact to the reference in both of its numerics.
#include "qattn_common.h"

namespace qattn {

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
    unsigned* amax_part;  // [G][kMomentSplits] workspace (head-wise): fp32 bits of the abs-max of every abs-max-pass block's share
    const unsigned* amax_ext;  // or: [G] fp32 bits of every head's abs-max, supplied by the caller (then amax_part is not used)
    int G, S, layout, token;
    // head-wise q and k of the fused step (else nullptr): every block of the abs-max pass leaves the sum of squares of its
    // share of the head in part[g][block]; the attention kernel adds a head's partial sums in a fixed order (deterministic, no
    // atomics, and no device-scope fence -- those write back a whole L2 on this chip: a last-block-reduces variant took
    // 517 us instead of 78) for its score-spread estimate (qattn_attn.h: predicted_r).
    float* part;          // [G][kMomentSplits]
    // where head g = (b, h) of the input starts and how far its rows are apart, in 16-byte vectors (a strided view of [B,H,S,D] with D
    // innermost and dense; dense: sb = H S D / 8, sh = S D / 8, ss = D / 8)
    int H;
    long sb, sh, ss;
};
__device__ __forceinline__ const uint4* job_head(const QuantJob& jb, int g) { return jb.x + (long)(g / jb.H) * jb.sb + (long)(g % jb.H) * jb.sh; }
struct QuantJobs {
    QuantJob j[3];
    unsigned* vexp;       // block-scaled V (else nullptr): [G of v][kMomentSplits] E8M0 bytes, one per 64-key chunk
    int nsplit;           // abs-max-pass blocks per head = valid entries of amax_part / part per head
    int zmap[3];          // abs-max pass: blockIdx.z -> job (the tensors that still need the pass)
    unsigned* zero_words; // quantise pass: zero_n words that its blocks clear between them for the kernel that follows (else nullptr)
    int zero_n;
};

// SV (both pre-pass kernels): false = every tensor of the launch is a dense [G,S,D] array -- the code these kernels had before strided views
// existed (compile-time row sizes; with the strides read at run time the dense C2 pre-pass measured +1.5 .. 2 us of 114:
// profiles/r06/kstats_c2_runtime_strides_in_prepass_vs_prev.log; as built now: equal, kstats_prepass_new_vs_prev.log); true = heads and rows
// through QuantJob::sb / sh / ss.  The launchers pick by the call.
template <int IN_FMT, bool SV>
__global__ __launch_bounds__(256) void amax_multi_kernel(const QuantJobs jobs, int D, int splits, int zbase) {
    const QuantJob& jb = jobs.j[jobs.zmap[zbase + blockIdx.z]];
    if (jb.token || (int)blockIdx.y >= jb.G) return;
    const long vecs_per_group = (long)jb.S * D / 8;
    const long g = blockIdx.y;
    const uint4* xg = SV ? job_head(jb, (int)g) : jb.x + g * vecs_per_group;
    // SV: vector i of the head = 16-byte piece (i mod D/8) of row i / (D/8); rows jb.ss vectors apart
    const int lg = 31 - __builtin_clz((unsigned)D >> 3);
    const long row_gap = SV ? jb.ss - (D >> 3) : 0L;   // (uniform; 0 for a dense head)
    const long per = (vecs_per_group + splits - 1) / splits;
    const long beg = (long)blockIdx.x * per;
    long end = beg + per;
    if (end > vecs_per_group) end = vecs_per_group;
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    unsigned m0 = 0, m1 = 0;
    f2 ss = {0.0f, 0.0f};
    const bool moments = jb.part != nullptr;   // (uniform per blockIdx.z)
    auto fold = [&](const uint4& v) {
        unsigned a = v.x & 0x7fff7fffu, b = v.y & 0x7fff7fffu, c = v.z & 0x7fff7fffu, d = v.w & 0x7fff7fffu;
        u16x2 pa, pb, pc, pd, p0, p1;
        __builtin_memcpy(&pa, &a, 4); __builtin_memcpy(&pb, &b, 4); __builtin_memcpy(&pc, &c, 4); __builtin_memcpy(&pd, &d, 4);
        __builtin_memcpy(&p0, &m0, 4); __builtin_memcpy(&p1, &m1, 4);
        p0 = __builtin_elementwise_max(p0, __builtin_elementwise_max(pa, pb));
        p1 = __builtin_elementwise_max(p1, __builtin_elementwise_max(pc, pd));
        __builtin_memcpy(&m0, &p0, 4); __builtin_memcpy(&m1, &p1, 4);
        if (moments) {
            // sum of squares, two elements per instruction: v_dot2c_f32_{bf16,f16} (fp32 accumulate).  The unpack + v_pk_fma_f32 form
            // took 12 VALU per 16 bytes and made this HBM-bound pass VALU-bound when the moments are on (AUTO); the sums only
            // feed the variance forecast (predicted_r), whose decision does not hang on the last bits.
            typedef __bf16 b2 __attribute__((ext_vector_type(2)));
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (IN_FMT == QATTN_FMT_BF16) {
                    b2 x;
                    __builtin_memcpy(&x, &w[i], 4);
                    ss[i & 1] = __builtin_amdgcn_fdot2_f32_bf16(x, x, ss[i & 1], false);
                } else {
                    h2 x;
                    __builtin_memcpy(&x, &w[i], 4);
                    ss[i & 1] = __builtin_amdgcn_fdot2(x, x, ss[i & 1], false);
                }
            }
        }
    };
    // kAmaxInFlight independent 16-byte loads in flight per thread (a plain strided loop kept ~2 and ran at 4.8 TB/s)
    auto at = [&](long i) { return SV ? xg + i + (i >> lg) * row_gap : xg + i; };   // (a strided view: every thread still folds the same elements)
    long i = beg + threadIdx.x;
    for (; i + (kAmaxInFlight - 1) * 256 < end; i += kAmaxInFlight * 256) {
        uint4 v[kAmaxInFlight];
#pragma unroll
        for (int u = 0; u < kAmaxInFlight; u++) v[u] = load_nt(at(i + u * 256));
#pragma unroll
        for (int u = 0; u < kAmaxInFlight; u++) fold(v[u]);
    }
    for (; i < end; i += 256) fold(*at(i));
    unsigned m = max(max(m0 & 0xffffu, m0 >> 16), max(m1 & 0xffffu, m1 >> 16));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    __shared__ unsigned red[4];
    __shared__ float red_ss[4];
    float s1 = ss.x + ss.y;
    if (moments) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s1 += __shfl_xor(s1, off);
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = m; red_ss[threadIdx.x >> 6] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        // no atomics and nothing to zero beforehand: the consumers (quantise pass, attention prologue) take the maximum of a
        // head's `splits` entries themselves
        m = max(max(red[0], red[1]), max(red[2], red[3]));
        jb.amax_part[g * kMomentSplits + blockIdx.x] = __float_as_uint(load16f<IN_FMT>((unsigned short)m));
        if (moments) jb.part[g * kMomentSplits + blockIdx.x] = (red_ss[0] + red_ss[1]) + (red_ss[2] + red_ss[3]);
    }
}

#ifndef QATTN_QUANT_TPB
#define QATTN_QUANT_TPB 1
#endif
constexpr int kQuantTilesPerBlock = QATTN_QUANT_TPB;   // 64-row tiles per block of the quantise pass (tuning knob, tools/bin variants)

template <int D, int IN_FMT, int OUT_FMT, bool SV>
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
    // a block takes kQuantTilesPerBlock consecutive tiles (downwards); the next tile's rows are requested before the current one is
    // converted, so a block has loads in flight all the time instead of one latency-bound burst per 16 KiB
    if (jobs.zero_words) {   // (every block takes a stride of them: 32 bytes of hand-out counters, or those and the peaked-group flags)
        const int nb = (int)(gridDim.x * gridDim.y * gridDim.z), lin = (int)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
        for (int i = lin * (int)blockDim.x + tid; i < jobs.zero_n; i += nb * (int)blockDim.x) jobs.zero_words[i] = 0u;
    }
    const int g = jb.G - 1 - (int)blockIdx.y, tile_first = (S + 63) / 64 - 1 - (int)blockIdx.x * kQuantTilesPerBlock;
    if (g < 0 || tile_first < 0) return;
    const int layout = jb.layout;
    const bool token = jb.token != 0;
    const float inv_qmax = (float)(1.0 / (double)(OUT_FMT == QATTN_FMT_E4M3 ? 448.0 : 57344.0));
    float scale = 1.0f;
    if (!token && !(layout == QATTN_LAYOUT_VFRAG && jobs.vexp != nullptr)) {
        const unsigned amax_bits = jb.amax_ext ? (jb.amax_ext[g] & 0x7fffffffu) : max_partials(jb.amax_part + (long)g * kMomentSplits, jobs.nsplit, tid & 63);
        scale = make_scale(__uint_as_float(amax_bits), inv_qmax, numerics, IN_FMT);
        if (tile_first - kQuantTilesPerBlock < 0 && tid == 0) jb.scale[g] = scale;   // (the block that holds tile 0)
    }
    float rinv = 1.0f / scale;
    const uint4* xg = SV ? job_head(jb, g) : jb.x + (long)g * S * VPR;
    const int row_vecs = SV ? (int)jb.ss : VPR;   // 16-byte vectors between consecutive rows
    const long Sp = (long)((S + 63) / 64) * 64;
    // block-scaled V (vblock_exponent, qattn_common.h): the tile IS the 64-key chunk; its rows are read once, reduced to the
    // chunk's abs-max through LDS, and quantised with the power-of-two scale that the attention kernel gets as one byte
    const bool vblock = layout == QATTN_LAYOUT_VFRAG && jobs.vexp != nullptr;   // uniform over the launch's z slice
    auto load_tile = [&](int tile, uint4 (&dst)[ITERS]) {
        const uint4* xt = xg + (long)tile * 64 * row_vecs;   // (block-uniform; a row's offset within the tile fits 32 bits: row stride <= 2^23 elements)
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const int vec = it * 256 + tid;
            const int r = vec / VPR, row = tile * 64 + r;
            dst[it] = make_uint4(0, 0, 0, 0);
            if (SV) { if (row < S) dst[it] = load_nt(&xt[(unsigned)r * (unsigned)row_vecs + (unsigned)(vec % VPR)]); }
            else if (row < S) dst[it] = load_nt(&xg[(long)row * VPR + vec % VPR]);
        }
    };
    uint4 held[ITERS];
    load_tile(tile_first, held);
#pragma unroll
    for (int ti = 0; ti < kQuantTilesPerBlock; ti++) {
        const int tile = tile_first - ti, row0 = tile * 64;
        if (tile < 0) break;   // (block-uniform)
        uint4 ahead[ITERS];
        const bool more = ti + 1 < kQuantTilesPerBlock && tile > 0;
        if (more) load_tile(tile - 1, ahead);
        bool exact_tile = false;   // block-scaled V: the chunk holds an inf or a NaN (scale 2^0 all the same) -> NaN bytes must survive
        if (vblock) {
            typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
            unsigned m0 = 0;
#pragma unroll
            for (int it = 0; it < ITERS; it++) {
                const unsigned w[4] = {held[it].x & 0x7fff7fffu, held[it].y & 0x7fff7fffu, held[it].z & 0x7fff7fffu, held[it].w & 0x7fff7fffu};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    u16x2 a, b;
                    __builtin_memcpy(&a, &w[i], 4); __builtin_memcpy(&b, &m0, 4);
                    b = __builtin_elementwise_max(a, b);
                    __builtin_memcpy(&m0, &b, 4);
                }
            }
            unsigned m = wave_allmax_u32(max(m0 & 0xffffu, m0 >> 16));
            unsigned* red = reinterpret_cast<unsigned*>(img);
            if ((tid & 63) == 0) red[tid >> 6] = m;
            __syncthreads();
            m = max(max(red[0], red[1]), max(red[2], red[3]));
            __syncthreads();   // (img is written below)
            const unsigned amax_bits = __float_as_uint(load16f<IN_FMT>((unsigned short)m));
            const int e = vblock_exponent(amax_bits, OUT_FMT);
            exact_tile = (amax_bits & 0x7f800000u) == 0x7f800000u;   // workgroup-uniform
            scale = __uint_as_float((unsigned)(e + 127) << 23);
            rinv = __uint_as_float((unsigned)(127 - e) << 23);
            if (tid == 0) {
                jobs.vexp[(long)g * kMomentSplits + tile] = (unsigned)(e + 127);
                if (tile == 0) jb.scale[g] = 1.0f;
            }
        }
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const int vec = it * 256 + tid;
            const int r = vec / VPR, dv = vec % VPR;
            const int row = row0 + r;
            const uint4 raw = held[it];
            if (token) {
                unsigned short e[8];
                __builtin_memcpy(e, &raw, 16);
                float a = 0.0f;
                bool nan = false;
#pragma unroll
                for (int j = 0; j < 8; j++) { const float f = load16f<IN_FMT>(e[j]); a = fmaxf(a, fabsf(f)); nan |= (f != f); }
                unsigned ab = nan ? 0x7fc00000u : __float_as_uint(a);
#pragma unroll
                for (int off = VPR / 2; off > 0; off >>= 1) ab = max(ab, (unsigned)__shfl_xor((int)ab, off));
                scale = make_scale(__uint_as_float(ab), inv_qmax, numerics, IN_FMT);
                if (dv == 0 && row < S) jb.scale[(long)g * S + row] = scale;
                rinv = 1.0f / scale;
            }
            const int2 lohi = quant8<IN_FMT, OUT_FMT>(raw, scale, rinv, exact_tile);
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
        if (layout != QATTN_LAYOUT_ROWMAJOR) {
            __syncthreads();
            if (layout == QATTN_LAYOUT_KFRAG) {
                uint4* og = jb.out + ((long)g * Sp + row0) * (D / 16);
                for (int i = tid; i < 64 * D / 16; i += 256) og[i] = *reinterpret_cast<const uint4*>(img + i * 16 + ((i >> 5) << 4));
            } else {
                vfrag_copy_out<D, VSTRIDE>(img, reinterpret_cast<unsigned char*>(jb.out) + ((long)g * Sp + row0) * D, tid);
            }
            if (more) __syncthreads();   // the image is free for the next tile
        }
        if (more) {
#pragma unroll
            for (int it = 0; it < ITERS; it++) held[it] = ahead[it];
        }
    }
}

// (Round 2's one-read experiment -- slices of a head held in LDS by persistent TEAMS of workgroups, abs-max exchanged through tagged
// granules in global memory: bit-exact, 6 % faster on K and V, 1 % on the step, speed dependent on whole teams being co-resident -- lived here
// behind QATTN_DEV until round 5; measurements in profiles/r02_oneread.md, code in the history up to commit 16f0f18.)

// ---------------------------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------------------------
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
    // head-wise: kMomentSplits abs-max words per head (one per block of the abs-max pass; nothing to zero beforehand)
    return scale_mode == QATTN_SCALE_HEAD ? (size_t)B * H * qattn::kMomentSplits * sizeof(unsigned) : 0;
}

template <int D>
static int launch_quant_multi(const QuantJobs& jobs, int in_fmt, int out_fmt, int numerics, dim3 grid, int ztop, hipStream_t st, bool sv = false);

// One tensor through the kernels of the fused q/k/v pre-pass (amax_multi_kernel: packed-u16 max, 8 loads in flight, one word per
// block and no atomics; quant_multi_kernel: conflict-free LDS images): the single-tensor entry used to have kernels of its own,
// an older abs-max (2 loads in flight, memset + atomicMax) that ran at 3.4 TB/s against 5.4.
extern "C" int qattn_quant_fp8(const void* x, int in_fmt, void* x8, float* scale, int B, int H, int S, int D,
                               int out_fmt, int scale_mode, int numerics, int out_layout, void* workspace,
                               size_t workspace_bytes, void* stream) {
    if (!x || !x8 || !scale || B <= 0 || H <= 0 || S <= 0) return QATTN_ERR_INVALID_ARG;
    if (D != 64 && D != 128 && D != 256) return QATTN_ERR_UNSUPPORTED_DIM;
    if (scale_mode != QATTN_SCALE_HEAD && scale_mode != QATTN_SCALE_TOKEN) return QATTN_ERR_INVALID_ARG;
    if (numerics != QATTN_NUMERICS_COMPILED && numerics != QATTN_NUMERICS_EAGER) return QATTN_ERR_INVALID_ARG;
    if (in_fmt != QATTN_FMT_BF16 && in_fmt != QATTN_FMT_FP16) return QATTN_ERR_UNSUPPORTED_FMT;
    if (out_fmt != QATTN_FMT_E4M3 && out_fmt != QATTN_FMT_E5M2) return QATTN_ERR_UNSUPPORTED_FMT;
    if (out_layout != QATTN_LAYOUT_ROWMAJOR && out_layout != QATTN_LAYOUT_KFRAG && out_layout != QATTN_LAYOUT_VFRAG) return QATTN_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int G = B * H;
    const bool head = scale_mode == QATTN_SCALE_HEAD;
    if (head && (!workspace || workspace_bytes < qattn_quant_workspace_bytes(B, H, S, D, scale_mode))) return QATTN_ERR_WORKSPACE;
    QuantJobs jobs;
    jobs.vexp = nullptr;
    jobs.zero_words = nullptr; jobs.zero_n = 0;
    jobs.nsplit = amax_splits(S, S, D);
    jobs.j[0] = QuantJob{(const uint4*)x, (uint4*)x8, scale, (unsigned*)workspace, nullptr, G, S, out_layout, head ? 0 : 1, nullptr,
                         H, (long)H * S * (D / 8), (long)S * (D / 8), D / 8};
    jobs.j[1] = jobs.j[2] = jobs.j[0];
    jobs.zmap[0] = jobs.zmap[1] = jobs.zmap[2] = 0;
    if (head) {
        dim3 grid(jobs.nsplit, G, 1), block(256);
        if (in_fmt == QATTN_FMT_BF16) hipLaunchKernelGGL((amax_multi_kernel<QATTN_FMT_BF16, false>), grid, block, 0, st, jobs, D, jobs.nsplit, 0);
        else hipLaunchKernelGGL((amax_multi_kernel<QATTN_FMT_FP16, false>), grid, block, 0, st, jobs, D, jobs.nsplit, 0);
    }
    dim3 grid(((S + 63) / 64 + kQuantTilesPerBlock - 1) / kQuantTilesPerBlock, G, 1);
    int rc;
    if (D == 64) rc = launch_quant_multi<64>(jobs, in_fmt, out_fmt, numerics, grid, 0, st);
    else if (D == 128) rc = launch_quant_multi<128>(jobs, in_fmt, out_fmt, numerics, grid, 0, st);
    else rc = launch_quant_multi<256>(jobs, in_fmt, out_fmt, numerics, grid, 0, st);
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

extern "C" int qattn_vblock_exponent(unsigned amax_bits, int out_fmt) { return qattn::vblock_exponent(amax_bits, out_fmt); }

extern "C" size_t qattn_quant_qkv_workspace_bytes(int B, int Hq, int Hkv) {
    if (B <= 0 || Hq <= 0 || Hkv <= 0) return 0;
    // [per-block abs-max bits of q, k, v | per-block sums of squares of q, k]: kMomentSplits words per head each; nothing to zero
    const size_t nq = (size_t)B * Hq, nk = (size_t)B * Hkv;
    return qattn::kMomentSplits * ((nq + 2 * nk) + (nq + nk)) * sizeof(unsigned);
}


template <int D>
static int launch_quant_multi(const QuantJobs& jobs, int in_fmt, int out_fmt, int numerics, dim3 grid, int ztop, hipStream_t st, bool sv) {
    dim3 block(256);
#define QATTN_QM(IN, OUT)                                                                                                              \
    do {                                                                                                                               \
        if (sv) hipLaunchKernelGGL((quant_multi_kernel<D, IN, OUT, true>), grid, block, 0, st, jobs, numerics, ztop);                  \
        else hipLaunchKernelGGL((quant_multi_kernel<D, IN, OUT, false>), grid, block, 0, st, jobs, numerics, ztop);                    \
    } while (0)
    if (in_fmt == QATTN_FMT_BF16 && out_fmt == QATTN_FMT_E4M3) QATTN_QM(QATTN_FMT_BF16, QATTN_FMT_E4M3);
    else if (in_fmt == QATTN_FMT_BF16 && out_fmt == QATTN_FMT_E5M2) QATTN_QM(QATTN_FMT_BF16, QATTN_FMT_E5M2);
    else if (in_fmt == QATTN_FMT_FP16 && out_fmt == QATTN_FMT_E4M3) QATTN_QM(QATTN_FMT_FP16, QATTN_FMT_E4M3);
    else if (in_fmt == QATTN_FMT_FP16 && out_fmt == QATTN_FMT_E5M2) QATTN_QM(QATTN_FMT_FP16, QATTN_FMT_E5M2);
    else return QATTN_ERR_UNSUPPORTED_FMT;
#undef QATTN_QM
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
                                   numerics, (unsigned*)workspace, false, false, false, (hipStream_t)stream);
}

int qattn::launch_quant_qkv(const void* q, const void* k, const void* v, int in_fmt, void* q8, void* k8, void* v8, float* scale_q,
                            float* scale_k, float* scale_v, int B, int Hq, int Hkv, int Sq, int Skv, int D, int out_fmt,
                            int scale_mode, int numerics, unsigned* ws, bool skip_q_payload, bool want_moments, bool v_block, hipStream_t st,
                            const float* const* ext_amax, unsigned* zero_words, int zero_n, const long long* strides) {
    const int tok = scale_mode == QATTN_SCALE_TOKEN;
    const unsigned* ext[3] = {nullptr, nullptr, nullptr};
    if (ext_amax)
        for (int t = 0; t < 3; t++) ext[t] = (tok && t < 2) ? nullptr : reinterpret_cast<const unsigned*>(ext_amax[t]);
    const size_t nq = (size_t)B * Hq, nk = (size_t)B * Hkv;
    float* part = reinterpret_cast<float*>(ws + kMomentSplits * (nq + 2 * nk));   // q heads then k heads
    const bool moments = !tok && want_moments;
    QuantJobs jobs;
    jobs.nsplit = amax_splits(Sq, Skv, D);
    const bool vblock = v_block && !tok && (Skv + 63) / 64 <= kMomentSplits;   // V's 256 words per head hold the chunks' scale bytes instead of abs-max words
    jobs.vexp = vblock ? ws + kMomentSplits * (nq + nk) : nullptr;
    jobs.zero_words = zero_words; jobs.zero_n = zero_n;
    // (strides: element strides {batch, head, row} of q, k, v -> 16-byte vectors; nullptr: dense)
    const long vd = D / 8;
    auto sv3 = [&](int t, int H, int S, int i) { return strides ? (long)(strides[3 * t + i] / 8) : i == 0 ? (long)H * S * vd : i == 1 ? (long)S * vd : vd; };
    jobs.j[0] = QuantJob{(const uint4*)q, (uint4*)q8, scale_q, ws, ext[0], B * Hq, Sq, QATTN_LAYOUT_ROWMAJOR, tok,
                         moments ? part : nullptr, Hq, sv3(0, Hq, Sq, 0), sv3(0, Hq, Sq, 1), sv3(0, Hq, Sq, 2)};
    jobs.j[1] = QuantJob{(const uint4*)k, (uint4*)k8, scale_k, ws + kMomentSplits * nq, ext[1], B * Hkv, Skv, QATTN_LAYOUT_KFRAG, tok,
                         moments ? part + nq * kMomentSplits : nullptr, Hkv, sv3(1, Hkv, Skv, 0), sv3(1, Hkv, Skv, 1), sv3(1, Hkv, Skv, 2)};
    jobs.j[2] = QuantJob{(const uint4*)v, (uint4*)v8, scale_v, ws + kMomentSplits * (nq + nk), ext[2], B * Hkv, Skv, QATTN_LAYOUT_VFRAG, 0, nullptr,
                         Hkv, sv3(2, Hkv, Skv, 0), sv3(2, Hkv, Skv, 1), sv3(2, Hkv, Skv, 2)};
    // the tensors the abs-max pass still has to read: head-wise ones without a caller-supplied abs-max (q, k: not with token-wise
    // scales; V always has one scale per head, or none of its own when block-scaled)
    int npass = 0;
    for (int t = 0; t < 3; t++)
        if ((!ext[t] || (moments && t < 2)) && (t == 2 ? !vblock : !tok)) jobs.zmap[npass++] = t;   // (moments: q and k are read for their sums of squares whoever supplies the abs-max)
    for (int t = npass; t < 3; t++) jobs.zmap[t] = 0;
    // (Tried and dropped: one tensor at a time -- amax then quantise, hoping the re-read hits the 256 MiB Infinity Cache --
    // was 13 % slower than the two fused launches; a one-pass register-resident variant with a cross-workgroup amax
    // exchange was 2-6x slower, the agent-scope atomics + spinning cost more than the second read.  A third variant --
    // all slices of a head pinned to ONE XCD, abs-max exchanged through that XCD's L2, slice re-read instead of held in
    // registers -- measured 0.42 ms vs 0.18: the exchange alone costs 0.2 ms and the re-read is NOT served on-die even with
    // residency capped to 2 workgroups per CU (amax only 0.071 ms, amax + re-read without any exchange 0.19 ms).
    // Round 2: slices held in LDS by persistent teams of workgroups (quant_team_kernel above, dev library): 6 % faster on
    // K and V, 1 % on the step; not shipped, see profiles/r02_oneread.md.  Non-temporal payload stores: the pre-pass
    // gains 2 % and the step loses 1 % -- the attention kernel finds the fresh fp8 K and V in the Infinity Cache.)
    const int Gmax = B * (Hq > Hkv ? Hq : Hkv), Smax = Sq > Skv ? Sq : Skv;
    {
        const int splits = amax_splits(Sq, Skv, D);
        dim3 grid(splits, Gmax, npass), block(256);   // (block-scaled V, supplied abs-max: not in the pass)
        if (npass > 0) {
            if (in_fmt == QATTN_FMT_BF16) {
                if (strides) hipLaunchKernelGGL((amax_multi_kernel<QATTN_FMT_BF16, true>), grid, block, 0, st, jobs, D, splits, 0);
                else hipLaunchKernelGGL((amax_multi_kernel<QATTN_FMT_BF16, false>), grid, block, 0, st, jobs, D, splits, 0);
            } else {
                if (strides) hipLaunchKernelGGL((amax_multi_kernel<QATTN_FMT_FP16, true>), grid, block, 0, st, jobs, D, splits, 0);
                else hipLaunchKernelGGL((amax_multi_kernel<QATTN_FMT_FP16, false>), grid, block, 0, st, jobs, D, splits, 0);
            }
        }
    }
    // the quantise pass walks jobs ztop, ztop-1, ...: with skip_q_payload only v and k (blockIdx.z = 0, 1)
    dim3 grid((((skip_q_payload ? Skv : Smax) + 63) / 64 + kQuantTilesPerBlock - 1) / kQuantTilesPerBlock, skip_q_payload ? B * Hkv : Gmax, skip_q_payload ? 2 : 3);
    int rc;
    const bool sv = strides != nullptr;
    if (D == 64) rc = launch_quant_multi<64>(jobs, in_fmt, out_fmt, numerics, grid, 2, st, sv);
    else if (D == 128) rc = launch_quant_multi<128>(jobs, in_fmt, out_fmt, numerics, grid, 2, st, sv);
    else rc = launch_quant_multi<256>(jobs, in_fmt, out_fmt, numerics, grid, 2, st, sv);
    if (rc != QATTN_OK) return rc;
    return hipGetLastError() == hipSuccess ? QATTN_OK : QATTN_ERR_LAUNCH;
}
