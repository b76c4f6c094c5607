// qattn_attn.h -- parameters and helpers shared by the attention kernel variants (gfx950 only).
#pragma once
#include "qattn_common.h"

namespace qattn {

constexpr int kWaves = 8;                    // waves per workgroup (2 per SIMD)
constexpr int kThreads = kWaves * 64;
constexpr int kQPerWave = 32;                // query rows per wave (MFMA N)
constexpr int kQPerWG = kWaves * kQPerWave;  // 256
constexpr float kPShift = 5.0f;              // P' = P * 2^5 keeps small probabilities above the e4m3 subnormals
constexpr float kRescaleThr = 3.0f;          // log2 units: P' <= 2^(5+3) = 256 < 448 (e4m3 max)
constexpr float kPShiftByte = 5.0f;          // byte-exponential mode: P' = P * 2^5 and a deferred-rescale threshold of 3:
constexpr float kRescaleThrByte = 3.0f;      //   P' <= 2^8 -> byte <= 120 < 0x7e; a tighter threshold (1) made the fix-up frequent
constexpr float kByteBias = -0.3f;           // centres the (1+m/8 >= 2^(m/8)) mantissa error of the byte exponential
constexpr int kTwoTermKeys = 1024;           // query blocks that see fewer keys than this use hi+lo (two-term) fp8 P from the start
// One-term rows are re-done with two-term P when R = l / p_max (the inverse of the row's largest softmax weight) ends below
// this: the error a single e4m3-rounded weight w contributes is about w * 2^-4 * |v - O|  
constexpr float kPeakR0 = 24.0f;

struct AttnParams {
    const unsigned char* q;
    const unsigned char* k;
    const unsigned char* v;
    void* out;
    float* lse;
    const float* sq;
    const float* sk;
    const float* sv;
    int B, Hq, Hkv, Sq, Skv;
    int nqb;         // query blocks (of kQPerWG rows) per head
    int nchunks;     // 64-key chunks per head
    int out_fmt;
    int xcd_remap;   // 1: each XCD gets a contiguous range of heads
    float sm_log2e;  // sm_scale * log2(e)
    int exact_exp;   // 1: v_exp_f32 + RNE fp8 conversion everywhere (no byte-exponential fast path)
    int precision;   // QATTN_PRECISION_*
    int two_term_keys;  // kTwoTermKeys (a development switch can change it)
    int n_two;       // leading query blocks per head that start in two-term mode (set by the launcher)
    float peak_r0;   // > 0: one-term blocks with a row of R < peak_r0 are repeated in two-term mode (QATTN_PRECISION_AUTO)
    unsigned* flags; // templated kernel (qattn_attn_v4.hip): one word per (head, 256-row block), set by the one-term launch
    long lse_stride; // floats between the LSE rows of consecutive (b, h)
    float lse_mul;   // 1 (natural log-sum-exp) or -sqrt(D) (QATTN_LSE_REFERENCE)
    const unsigned char* q16;      // fused step: the 16-bit (bf16) Q tensor, quantised row by row in the kernel prologue (else nullptr)
    const unsigned* q_amax_bits;   // fused step: per-(b,h) abs-max bits of Q from the amax pass
    float* sq_out;                 // fused step: scale_q [B,Hq] is written by the attention kernel
    int q_numerics;
#ifdef QATTN_DEV
    int waves;       // waves per workgroup of the v2 kernel (8 or 4): nqb is computed for waves*32 rows
    int lds_pad;     // force this dynamic-LDS size (occupancy experiments), 0 = natural
    unsigned long long* dbg_buf;  // per-wave {cycles, realtime ticks} of the KV sweep when dbg & 16
    int dbg;         // 16 = stamp per-wave sweep cycles into dbg_buf
#endif
};

template <int CBSZ, int BLGP>
__device__ inline v16f mfma_f8(v8i a, v8i b, v16f c) {
    // scale operands 0 -> the unscaled v_mfma_f32_32x32x64_f8f6f4 (implicit scale 1.0; profiles/r01_mfma_probe.log)
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, CBSZ, BLGP, 0, 0, 0, 0);
}

// two floats -> two bf16 / fp16 in one dword (round to nearest even; v_cvt_pk_bf16_f32 / v_cvt_pkrtz is NOT used for fp16)
__device__ inline unsigned pack2_bf16(float a, float b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const b2 h = __builtin_convertvector(f2{a, b}, b2);
    unsigned u;
    __builtin_memcpy(&u, &h, 4);
    return u;
}
__device__ inline unsigned pack2_f16(float a, float b) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 h = {(_Float16)a, (_Float16)b};
    unsigned u;
    __builtin_memcpy(&u, &h, 4);
    return u;
}

__device__ inline v8i lds_read_frag(const unsigned char* base) {
    // two ds_read_b128: pieces [half=0] and [half=1] are 512 bytes apart
    v4i lo = *reinterpret_cast<const v4i*>(base);
    v4i hi = *reinterpret_cast<const v4i*>(base + 512);
    v8i r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// Epilogue shared by the attention kernels: O^T accumulators (query on the lane, d in the registers) -> normalised 16-bit
// rows.  The lanes of the two half-waves hold the 8-byte halves of each 16-byte piece of a row; one v_permlane32_swap per
// dword pairs them up so that every lane stores 16 contiguous bytes (half the store instructions of the 8-byte form).
template <int MB>
__device__ __forceinline__ void store_o_rows(void* out, int out_fmt, const v16f (&o)[MB], float inv, long row, int hh, bool valid) {
    unsigned char* op = reinterpret_cast<unsigned char*>(out) + row * (MB * 32) * 2 + (hh << 4);
    const bool bf = out_fmt == QATTN_FMT_BF16;
#pragma unroll
    for (int m = 0; m < MB; m++)
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            unsigned a[2], b[2];  // this lane's 4 + 4 elements of column groups j and j + 1
#pragma unroll
            for (int i = 0; i < 2; i++) {
                a[i] = bf ? pack2_bf16(o[m][4 * j + 2 * i] * inv, o[m][4 * j + 2 * i + 1] * inv)
                          : pack2_f16(o[m][4 * j + 2 * i] * inv, o[m][4 * j + 2 * i + 1] * inv);
                b[i] = bf ? pack2_bf16(o[m][4 * j + 4 + 2 * i] * inv, o[m][4 * j + 5 + 2 * i] * inv)
                          : pack2_f16(o[m][4 * j + 4 + 2 * i] * inv, o[m][4 * j + 5 + 2 * i] * inv);
                const auto sw = __builtin_amdgcn_permlane32_swap(a[i], b[i], false, false);
                a[i] = sw[0]; b[i] = sw[1];
            }
            // lanes 0..31: [own group j | upper half's group j] = columns 32m + 8j .. +7; lanes 32..63: the next 8 columns
            // (the exchange runs with every lane active; only the store is predicated on the row being inside the tensor)
            if (valid) *reinterpret_cast<v4i*>(op + (32 * m + 8 * j) * 2) = v4i{(int)a[0], (int)a[1], (int)b[0], (int)b[1]};
        }
}

template <int N>
__device__ inline void wait_vmcnt() {
    static_assert(N == 0 || N == 1 || N == 2 || N == 4, "unsupported vmcnt");
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    if (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
}

// block -> (head, query block).  Blocks b and b+8 share an XCD (round-robin dispatch; a speed assumption only):
// each XCD gets a contiguous range of heads so the 1-2 heads it works on keep their K/V in its private 4 MiB L2.
__device__ inline void map_block(const AttnParams& p, int bid, int nqb, bool causal, int& head, int& qb) {
    if (p.xcd_remap) {
        const int xcd = bid & 7, idx = bid >> 3;
        head = xcd * ((p.B * p.Hq) >> 3) + idx / nqb;
        qb = idx % nqb;
    } else {
        head = bid / nqb;
        qb = bid % nqb;
    }
    if (causal) qb = nqb - 1 - qb;  // heaviest query blocks first
}

bool attn_v2_covers(int D, int causal, int scale_mode);
int launch_attn_v2(const AttnParams& p, int D, int fmt, int causal, int scale_mode, hipStream_t st);
int launch_attn_v4_full(const AttnParams& p, int D, int fmt, int causal, int scale_mode, hipStream_t st);


}  // namespace qattn
