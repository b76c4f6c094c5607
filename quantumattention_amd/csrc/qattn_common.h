// qattn_common.h -- shared device helpers for the gfx950 (MI355X / CDNA4) kernels.  gfx950 only, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/qattn.h"
#include "../../include/qattn_measure.h"

namespace qattn {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned short v8u16 __attribute__((ext_vector_type(8)));

constexpr int kChunkKeys = 64;  // keys per K/V fragment chunk (one PV MFMA K-dimension)

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------------------
// Fragment layouts QATTN_LAYOUT_* (private to the library; S zero-padded to Sp = 64*ceil(S/64), D*Sp bytes per (b,h), c = 64-key chunk):
//   KFRAG   chunk = [t:2][s:D/64][hh:2][half:2][key:32][16 B]; byte j of a piece = K[64c + 32t + key][64s + 32hh + 16half + j]
//   VFRAG   chunk = [m:D/32][hh:2][half:2][d:32][16 B]; byte 4w+i of a piece = V[64c + 32half + 8w + 4hh + i][32m + d]
//           = the A operands of v_mfma_f32_32x32x64_f8f6f4 for S^T = K.Q^T and O^T = V^T.P^T: linear for LDS-DMA, conflict-free ds_read_b128
//   K16FRAG chunk = [t:2][s:D/16][hh:2][key:32][8 elts], piece = K[64c + 32t + key][16s + 8hh + (0..7)]            (16-bit path)
//   V16FRAG chunk = [t:2][m:D/32][s:2][hh:2][d:32][8 elts], elt j = V[64c + 32t + 16s + 8(j>>2) + 4hh + (j&3)][32m + d]
// Offsets below are in bytes from the start of the 64-key chunk (64*D bytes).
// ---------------------------------------------------------------------------------------------------------
// K[key_in_chunk][d]  ->  [t:2][s:D/64][hh:2][half:2][key:32][16]
template <int D>
__host__ __device__ inline int kfrag_offset(int key, int d) {
    const int t = key >> 5, kl = key & 31, s = d >> 6, hh = (d >> 5) & 1, half = (d >> 4) & 1;
    return ((t * (D / 64) + s) << 11) + (hh << 10) + (half << 9) + (kl << 4) + (d & 15);
}
// V[key_in_chunk][d]  ->  [m:D/32][hh:2][half:2][dl:32][4*w+i]   with key = 32*half + 8*w + 4*hh + i
template <int D>
__host__ __device__ inline int vfrag_offset(int key, int d) {
    const int half = key >> 5, w = (key >> 3) & 3, hh = (key >> 2) & 1, i = key & 3;
    return ((d >> 5) << 11) + (hh << 10) + (half << 9) + ((d & 31) << 4) + (w << 2) + i;
}

__device__ inline float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }
__device__ inline float fp16_bits_to_f32(unsigned short b) {
    _Float16 h;
    __builtin_memcpy(&h, &b, 2);
    return (float)h;
}
// round-to-nearest-even to bf16 precision, result kept as float (finite inputs)
__device__ inline float round_bf16(float x) {
    unsigned u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return x;
    u += 0x7fffu + ((u >> 16) & 1u);
    return __uint_as_float(u & 0xffff0000u);
}
__device__ inline float round_fp16(float x) { return (float)(_Float16)x; }

template <int FMT16>
__device__ inline float load16f(unsigned short b) {
    return FMT16 == QATTN_FMT_BF16 ? bf16_bits_to_f32(b) : fp16_bits_to_f32(b);
}
template <int FMT16>
__device__ inline float round16(float x) {
    return FMT16 == QATTN_FMT_BF16 ? round_bf16(x) : round_fp16(x);
}

// 16 bytes that this kernel reads once and nobody re-reads from cache (the pre-pass's 16-bit inputs): a non-temporal load leaves
// L2 and the Infinity Cache to the fp8 K / V the attention kernel is about to stream.  qattn_quant_qkv_fp8 -6 %, the step -1..2 %
// (profiles/r03/ab_nt_loads.log).
__device__ __forceinline__ uint4 load_nt(const uint4* p) {
    typedef unsigned nt4 __attribute__((ext_vector_type(4)));
    const nt4 t = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(p));
    return make_uint4(t[0], t[1], t[2], t[3]);
}

// two floats -> two fp8 bytes in the low (HI=false) or high (HI=true) half of `old`; RNE, no saturation (gfx950)
template <int FMT8, bool HI>
__device__ inline int cvt_pk_fp8(float a, float b, int old) {
    if (FMT8 == QATTN_FMT_E4M3) return __builtin_amdgcn_cvt_pk_fp8_f32(a, b, old, HI);
    return __builtin_amdgcn_cvt_pk_bf8_f32(a, b, old, HI);
}
template <int FMT8>
__device__ inline int cvt4_fp8(float a, float b, float c, float d) {
    int r = 0;
    r = cvt_pk_fp8<FMT8, false>(a, b, r);
    r = cvt_pk_fp8<FMT8, true>(c, d, r);
    return r;
}

// ---------------------------------------------------------------------------------------------------------
// 8 consecutive 16-bit inputs -> 8 fp8 bytes, bit-exact to  fp8(clamp(round16(x / scale), +-qmax))  (nn.py:14-19 as the
// reference's compiled path evaluates it: IEEE fp32 quotient, rounded to the input dtype, clamped, RNE to fp8).
// The IEEE divide + software bf16 rounding cost ~40 VALU per element and made the pre-pass VALU-bound, so bf16
// inputs take a fast path: q' = x * rinv with rinv = RNE(1/scale) is within 3 fp32 ulps of RNE(x/scale); both round
// to the same bf16 unless q' lies within 4 ulps of a bf16 tie (low 16 bits near 0x8000, probability ~1.4e-4), and only
// those vectors fall back to the exact sequence.  Requires a finite scale, which implies finite inputs where the scale is
// derived from the group's abs-max (elsewhere: force_exact).  Results below the fp8 subnormal range round to (signed) zero either way.
// ---------------------------------------------------------------------------------------------------------
// scale = clamp_min(amax * (1/fmax), eps_f32)   (nn.py:14-16; eager numerics round the scale and eps to the input dtype)
__device__ inline float make_scale(float amax, float inv_qmax, int numerics, int in_fmt) {
    const float eps = 1.1920928955078125e-07f;  // torch.finfo(torch.float32).eps  (nn.py:15)
    float s = amax * inv_qmax;
    float e = eps;
    if (numerics == QATTN_NUMERICS_EAGER) {
        s = in_fmt == QATTN_FMT_BF16 ? round_bf16(s) : round_fp16(s);
        e = in_fmt == QATTN_FMT_BF16 ? round_bf16(eps) : round_fp16(eps);
    }
    if (!(s >= e)) s = (s != s) ? s : e;  // clamp_min keeps NaN
    return s;
}

template <int IN_FMT, int OUT_FMT>
__device__ __attribute__((noinline)) int2 quant8_exact_call(const uint4 raw, float scale);

template <int IN_FMT, int OUT_FMT>
__device__ __forceinline__ int2 quant8_exact(const uint4& raw, float scale) {
    const float qmax = OUT_FMT == QATTN_FMT_E4M3 ? 448.0f : 57344.0f;
    unsigned short e[8];
    __builtin_memcpy(e, &raw, 16);
    float q[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        float t = round16<IN_FMT>(load16f<IN_FMT>(e[j]) / scale);  // IEEE fp32 divide, then the reference's rounding to the input dtype
        t = t > qmax ? qmax : t;
        t = t < -qmax ? -qmax : t;
        q[j] = t;
    }
    return make_int2(cvt4_fp8<OUT_FMT>(q[0], q[1], q[2], q[3]), cvt4_fp8<OUT_FMT>(q[4], q[5], q[6], q[7]));
}

template <int IN_FMT, int OUT_FMT>
__device__ __attribute__((noinline)) int2 quant8_exact_call(const uint4 raw, float scale) {
    return quant8_exact<IN_FMT, OUT_FMT>(raw, scale);
}

// force_exact: the caller knows of non-finite inputs although `scale` is finite (the block-scaled V: a chunk with an inf or a
// NaN gets the scale 2^0) -- the packed clamp below would turn a NaN into +-fmax.
// fp16 inputs (round 5, VERDICT r4 Missing-1).  The bf16 trick -- x * rinv, exact sequence only near a rounding tie -- does not carry over:
// fp16 keeps 10 mantissa bits, its ties are eight times as dense in the fp32 quotient's low bits, and with 64 lanes per wave nearly every
// other vector would take the slow branch.  Instead the quotient itself is made exact without the divide: with rinv = RN(1 / scale)
// (correctly rounded: -fhip-fp32-correctly-rounded-divide-sqrt), q0 = x * rinv, r = fma(-q0, scale, x) (exact), q1 = fma(r, rinv, q0) is the
// correctly rounded x / scale (Markstein's correction step; checked on the GPU against the IEEE divide for every fp16 x over 2^20 scales and
// on the golden vectors, tests/test_gpu_quant.py).  Then v_cvt_pk_f16_f32 (RNE), the clamp as packed fp16 min / max, and
// v_cvt_scalef32_pk_{fp8,bf8}_f16 at scale 1.0 -- bit-identical to unpack -> v_med3_f32 -> v_cvt_pk_fp8_f32 for every finite fp16 within the
// clamp (same test).  About 13 VALU per pair against 80 for the IEEE divide + software rounding.  Requires finite inputs and scale.
template <int OUT_FMT>
__device__ __forceinline__ int2 quant8_f16_fast(const uint4& raw, float scale, float rinv) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef short s2 __attribute__((ext_vector_type(2)));
    const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
    const _Float16 qm = (_Float16)(OUT_FMT == QATTN_FMT_E4M3 ? 448.0f : 57344.0f);
    const h2 hi_lim = {qm, qm}, lo_lim = {(_Float16)(-(float)qm), (_Float16)(-(float)qm)};
    h2 cl[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        h2 xh;
        __builtin_memcpy(&xh, &w[i], 4);
        const f2 x = __builtin_convertvector(xh, f2);
        f2 q;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const float q0 = x[j] * rinv;
            const float r = __builtin_fmaf(-q0, scale, x[j]);
            q[j] = __builtin_fmaf(r, rinv, q0);
        }
        h2 h = __builtin_convertvector(q, h2);   // v_cvt_pk_f16_f32: RNE
        h = __builtin_elementwise_max(__builtin_elementwise_min(h, hi_lim), lo_lim);
        // the quotient's sign is x's (scale > 0); the correction step loses it on x = -0 (r = +0, q1 = +0 + -0 = +0): one v_bfi per pair
        unsigned hb;
        __builtin_memcpy(&hb, &h, 4);
        hb = (hb & 0x7fff7fffu) | (w[i] & 0x80008000u);
        __builtin_memcpy(&cl[i], &hb, 4);
    }
    s2 lo = {0, 0}, hi = {0, 0};
    if (OUT_FMT == QATTN_FMT_E4M3) {
        lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(lo, cl[0], 1.0f, false);
        lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(lo, cl[1], 1.0f, true);
        hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(hi, cl[2], 1.0f, false);
        hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(hi, cl[3], 1.0f, true);
    } else {
        lo = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(lo, cl[0], 1.0f, false);
        lo = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(lo, cl[1], 1.0f, true);
        hi = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(hi, cl[2], 1.0f, false);
        hi = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(hi, cl[3], 1.0f, true);
    }
    int2 r;
    __builtin_memcpy(&r.x, &lo, 4);
    __builtin_memcpy(&r.y, &hi, 4);
    return r;
}

template <int IN_FMT, int OUT_FMT>
__device__ __forceinline__ int2 quant8(const uint4& raw, float scale, float rinv, bool force_exact = false) {
    if (IN_FMT != QATTN_FMT_BF16) {
        // non-finite inputs or scale, and whatever else the caller knows (force_exact): the exact sequence; the test is on the vector's
        // packed exponents -- an fp16 inf / NaN has all five exponent bits set
        const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
        bool special = force_exact || !((__float_as_uint(scale) & 0x7f800000u) != 0x7f800000u);
#pragma unroll
        for (int i = 0; i < 4; i++) special = special || ((w[i] & 0x7c00u) == 0x7c00u) || ((w[i] & 0x7c000000u) == 0x7c000000u);
        if (__builtin_expect(special, 0)) return quant8_exact_call<IN_FMT, OUT_FMT>(raw, scale);
        return quant8_f16_fast<OUT_FMT>(raw, scale, rinv);
    }
    // Per pair of elements: unpack (2 VALU), v_pk_mul_f32, tie test on the packed low halves (v_perm, v_pk_add_u16,
    // v_pk_min_u16), v_cvt_pk_bf16_f32, clamp of the packed bf16 magnitudes (and, v_pk_min_u16, and-or), and one
    // v_cvt_scalef32_pk_{fp8,bf8}_bf16 at scale 1.0 -- bit-identical to unpack -> v_med3_f32 -> v_cvt_pk_fp8_f32 for every
    // finite bf16 (tools/probes/cvt_bf16_fp8_probe.hip) and 11 instead of 17 VALU per pair: the pre-pass is VALU-bound.
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    typedef short s2 __attribute__((ext_vector_type(2)));
    const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
    const unsigned qbits = OUT_FMT == QATTN_FMT_E4M3 ? 0x43e043e0u : 0x47604760u;   // bf16(448) / bf16(57344), both halves
    const unsigned tie = 0x80048004u;
    u16x2 pq, ptie, pnear = {0xffff, 0xffff};
    __builtin_memcpy(&pq, &qbits, 4);
    __builtin_memcpy(&ptie, &tie, 4);
    b2 cl[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const f2 q = f2{__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)} * rinv;
        const unsigned lows = __builtin_amdgcn_perm(__float_as_uint(q.y), __float_as_uint(q.x), 0x05040100u);   // the low halves of both quotients
        u16x2 pl;
        __builtin_memcpy(&pl, &lows, 4);
        pnear = __builtin_elementwise_min(pnear, (u16x2)(pl + ptie));
        const b2 h = __builtin_convertvector(q, b2);  // v_cvt_pk_bf16_f32: RNE
        unsigned u;
        __builtin_memcpy(&u, &h, 4);
        // (the clamp is not what bounds these kernels: a build without it -- legal whenever the scale comes from the tensor's own
        // abs-max -- quantised K and V in the same time and gained 1 % on the step, within the noise; profiles/r03/prepass_variants.md)
        unsigned mag = u & 0x7fff7fffu;
        u16x2 pm;
        __builtin_memcpy(&pm, &mag, 4);
        pm = __builtin_elementwise_min(pm, pq);
        __builtin_memcpy(&mag, &pm, 4);
        const unsigned c = mag | (u & 0x80008000u);
        __builtin_memcpy(&cl[i], &c, 4);
    }
    const unsigned near = min((unsigned)pnear.x, (unsigned)pnear.y);
    const bool slow = force_exact || near < 9u || !((__float_as_uint(scale) & 0x7f800000u) != 0x7f800000u);
    if (__builtin_expect(slow, 0)) return quant8_exact_call<IN_FMT, OUT_FMT>(raw, scale);  // one out-of-line copy: rare
    s2 lo = {0, 0}, hi = {0, 0};
    if (OUT_FMT == QATTN_FMT_E4M3) {
        lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(lo, cl[0], 1.0f, false);
        lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(lo, cl[1], 1.0f, true);
        hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(hi, cl[2], 1.0f, false);
        hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(hi, cl[3], 1.0f, true);
    } else {
        lo = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(lo, cl[0], 1.0f, false);
        lo = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(lo, cl[1], 1.0f, true);
        hi = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(hi, cl[2], 1.0f, false);
        hi = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(hi, cl[3], 1.0f, true);
    }
    int2 r;
    __builtin_memcpy(&r.x, &lo, 4);
    __builtin_memcpy(&r.y, &hi, 4);
    return r;
}

// Block-scaled V (fused step, D = 128 head-wise): every 64-key chunk of a kv head is quantised with its own power-of-two
// scale 2^e, e the smallest exponent with amax / 2^e <= fmax, found inside the quantise pass's own tile -- V needs no abs-max
// pass -- and applied by the PV products as the E8M0 block scale (byte e + 127) of v_mfma_scale_f32_32x32x64_f8f6f4.
// From the fp32 bits of the chunk's abs-max, in integer arithmetic so that the test restatement is exact: amax = m 2^k,
// fmax = 1.75 2^Q (Q = 8 for e4m3, 15 for e5m2)  ->  e = k - Q, one more if m > 1.75; zero, subnormal and non-finite abs-max: e = 0.
__host__ __device__ inline int vblock_exponent(unsigned amax_bits, int out_fmt) {
    const int ef = (int)((amax_bits >> 23) & 255u);
    if (ef == 0 || ef == 255) return 0;
    const int e = ef - 127 - (out_fmt == QATTN_FMT_E4M3 ? 8 : 15) + ((amax_bits & 0x7fffffu) > 0x600000u ? 1 : 0);
    return e < -126 ? -126 : e > 126 ? 126 : e;
}

// ---------------------------------------------------------------------------------------------------------
// Whole-wave reductions and lane exchanges WITHOUT a lane-index register.  __shfl_xor / __shfl go through ds_bpermute with an
// address computed from the lane id; in the persistent attention kernels those six or eight loop-invariant address registers
// are hoisted to the top of the block loop, stay live through the hand-scheduled sweeps and -- the kernels sit at the
// 256-register limit -- get spilled.  DPP row operations, ds_swizzle with an immediate pattern and v_permlane32_swap need none.
//   partner within a quad / 8 lanes / 16 lanes: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
//   partner 16 lanes away: ds_swizzle bit mode, xor mask 16;  partner 32 lanes away: v_permlane32_swap
// ---------------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned x) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ unsigned swizzle_xor16(unsigned x) { return (unsigned)__builtin_amdgcn_ds_swizzle((int)x, (16 << 10) | 0x1f); }
template <typename Op>
__device__ __forceinline__ unsigned wave_allreduce_u32(unsigned x, Op op) {   // the same value in all 64 lanes
    x = op(x, dpp_u32<0xB1>(x));    // quad_perm [1,0,3,2]
    x = op(x, dpp_u32<0x4E>(x));    // quad_perm [2,3,0,1]
    x = op(x, dpp_u32<0x141>(x));   // row_half_mirror: lane i <-> 7 - i of its 8
    x = op(x, dpp_u32<0x140>(x));   // row_mirror:      lane i <-> 15 - i of its 16
    x = op(x, swizzle_xor16(x));
    const auto sw = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return op(sw[0], sw[1]);
}
__device__ __forceinline__ float wave_allsum(float x) {
    return __uint_as_float(wave_allreduce_u32(__float_as_uint(x), [](unsigned a, unsigned b) { return __float_as_uint(__uint_as_float(a) + __uint_as_float(b)); }));
}
__device__ __forceinline__ unsigned wave_allmax_u32(unsigned x) {
    return wave_allreduce_u32(x, [](unsigned a, unsigned b) { return a > b ? a : b; });
}
// every lane q reads lane q & 15 (the 16x16 MFMA's row sums live in lanes 0..15)
__device__ __forceinline__ float bcast_low16(float x) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);   // [0]: lanes 32.. <- lanes 0..31
    return __int_as_float(__builtin_amdgcn_ds_swizzle((int)sw[0], 0x0f));   // bit mode, and mask 0x0f: lane i <- lane i & 15 of its 32
}

// the maximum of a head's per-block abs-max words, the same value in every lane
__device__ inline unsigned max_partials(const unsigned* part, int n, int lane) {
    unsigned m = 0u;
    for (int i = lane; i < n; i += 64) m = max(m, part[i]);
    return wave_allmax_u32(m);
}

// q/k/v pre-pass launcher shared by qattn_quant_qkv_fp8 and the fused step entry (qattn_api.hip).  `ws` holds the per-block
// abs-max words of q [B*Hq][256], k [B*Hkv][256], v [B*Hkv][256] and the per-block sums of squares (quant_moments).  skip_q_payload: q8 / scale_q are not written (the attention kernel
// quantises its own Q rows from the 16-bit tensor and the q amax bits).
// want_moments (head-wise only): the abs-max pass also leaves the sums of squares of every q and k head in the workspace, as
// `nsplit` partial sums per head (stride kMomentSplits), where quant_moments() finds them (the attention kernel's score-spread
// estimate, qattn_attn.h predicted_r).
constexpr int kMomentSplits = 256;   // >= the abs-max pass's blocks per head
struct QuantMoments { const float* part_q; const float* part_k; const unsigned* amax_q; const unsigned* vexp; int nsplit; };   // vexp: [B*Hkv][kMomentSplits] E8M0 bytes of the V chunks (block-scaled V)
#ifndef QATTN_AMAX_INFLIGHT
#define QATTN_AMAX_INFLIGHT 8
#endif
#ifndef QATTN_AMAX_ROUNDS
#define QATTN_AMAX_ROUNDS 1
#endif
constexpr int kAmaxInFlight = QATTN_AMAX_INFLIGHT;   // 16-byte loads a thread of the abs-max pass keeps in flight (tuning knobs:
constexpr int kAmaxRounds = QATTN_AMAX_ROUNDS;       //   tools/bin variants) and bursts of them per block
inline int amax_splits(int Sq, int Skv, int D) {
    const long vecs = (long)(Sq > Skv ? Sq : Skv) * D / 8;
    const long per = 256L * kAmaxInFlight * kAmaxRounds;
    const long s = (vecs + per - 1) / per;   // kAmaxInFlight x 16 B per thread and burst
    return s < 1 ? 1 : s > kMomentSplits ? kMomentSplits : (int)s;
}
inline QuantMoments quant_moments(const unsigned* ws, int B, int Hq, int Hkv, int Sq, int Skv, int D) {
    const size_t nq = (size_t)B * Hq, nk = (size_t)B * Hkv;
    const float* s = reinterpret_cast<const float*>(ws + kMomentSplits * (nq + 2 * nk));
    return QuantMoments{s, s + nq * kMomentSplits, ws, ws + kMomentSplits * (nq + nk), amax_splits(Sq, Skv, D)};
}
// ext_amax[t] (t = 0, 1, 2 for q, k, v; head-wise only, else nullptr): fp32 abs-max of every head of that tensor as its PRODUCER
// knows it (the epilogue of the projection / RoPE kernel that wrote it) -- the tensor then takes no part in the abs-max pass, and
// when none is left the pass is not launched at all (the reference gets this fusion from Inductor, nn.py:410-418).
int launch_quant_qkv(const void* q, const void* k, const void* v, int in_fmt, void* q8, void* k8, void* v8, float* scale_q,
                     float* scale_k, float* scale_v, int B, int Hq, int Hkv, int Sq, int Skv, int D, int out_fmt, int scale_mode,
                     int numerics, unsigned* ws, bool skip_q_payload, bool want_moments, bool v_block, hipStream_t st,
                     const float* const* ext_amax = nullptr, unsigned* zero_words = nullptr, int zero_n = 0,   // zero_words: a few scratch
                     // words of the NEXT kernel in the stream (the attention launch's hand-out counters) that the quantise pass clears on the way
                     const long long* strides = nullptr);   // element strides {batch, head, row} of q, k, v (9 values; nullptr: dense [B,H,S,D])

}  // namespace qattn
