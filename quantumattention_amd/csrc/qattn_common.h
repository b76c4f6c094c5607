// qattn_common.h -- shared device helpers for the gfx950 (MI355X / CDNA4) kernels.  gfx950 only, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/qattn.h"

namespace qattn {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned short v8u16 __attribute__((ext_vector_type(8)));

constexpr int kChunkKeys = 64;  // keys per K/V fragment chunk (one PV MFMA K-dimension)

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------------------
// Fragment layouts (include/qattn.h).  Offsets are in bytes from the start of the 64-key chunk (64*D bytes).
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

}  // namespace qattn
