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
constexpr int kTwoTermKeys = 1024;           // rows that see fewer keys than this use hi+lo (two-term) fp8 P

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
    int waves;       // waves per workgroup of the v2 kernel (8 or 4): nqb is computed for waves*32 rows
    int lds_pad;     // development: force this dynamic-LDS size (occupancy experiments), 0 = natural
    unsigned long long* dbg_buf;  // development: per-wave {cycles, realtime ticks} of the KV sweep when dbg & 16
    int dbg;         // development: 16 = stamp per-wave sweep cycles into dbg_buf
    const unsigned char* q16;      // fused step: the 16-bit (bf16) Q tensor, quantised row by row in the kernel prologue (else nullptr)
    const unsigned* q_amax_bits;   // fused step: per-(b,h) abs-max bits of Q from the amax pass
    float* sq_out;                 // fused step: scale_q [B,Hq] is written by the attention kernel
    int q_numerics;
    int use_v4;      // 1: head-wise one-term byte-exponential q-blocks run on the three-waves-per-SIMD kernel (qattn_attn_v4.hip)
};

template <int CBSZ, int BLGP>
__device__ inline v16f mfma_f8(v8i a, v8i b, v16f c) {
    // scale operands 0 -> the unscaled v_mfma_f32_32x32x64_f8f6f4 (implicit scale 1.0; profiles/r01_mfma_probe.log)
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, CBSZ, BLGP, 0, 0, 0, 0);
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

int launch_attn_v2(const AttnParams& p, int D, int fmt, int causal, int scale_mode, hipStream_t st);
int launch_attn_v3(const AttnParams& p, int D, int fmt, int causal, int scale_mode, hipStream_t st);
int launch_attn_v4(const AttnParams& p, int D, int fmt, int causal, int scale_mode, int row_lo, hipStream_t st);
int launch_attn_v4_full(const AttnParams& p, int D, int fmt, int causal, int scale_mode, hipStream_t st);

}  // namespace qattn
