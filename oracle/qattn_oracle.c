/*
 * qattn_oracle.c -- CPU ORACLE for the FP8 fused-attention hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing under quantumattention_amd/ (the product) may import, link or call this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * It restates, in plain C, the algorithm of the reference (WaveSpeedAI/QuantumAttention @ 2025-02-22):
 *
 *   qo_quantize_fp8      <- src/quantum_attn/nn.py:14-19  (_dynamically_quantize_fp8), in both numerics
 *                           the reference exhibits: "compiled" (nn.py:22-42 / nn.py:521-539, what its GPU
 *                           path runs, fp32 scale) and "eager" (all arithmetic in the input dtype).
 *   qo_attention_forward <- src/quantum_attn/ops.py:64-95 (_fp8_attention_forward: de-quantise, then
 *                           aten.scaled_dot_product_attention) and ops.py:17-29 (_attention_forward),
 *                           evaluated in fp64 on the SAME quantised inputs (SURVEY.md §8c oracle O2/O3).
 *   fp8 codecs           <- torch.float8_e4m3fn / float8_e5m2 (OCP FP8), round-to-nearest-even, as used by
 *                           nn.py:18 `.to(torch.float8_e4m3fn)`.
 *
 * Pinned (tests/test_oracle_golden.py) against the tests/golden npz fixtures, which tests/golden/gen_golden.py
 * generated in the build container by running the reference's own functions:
 *   - quantiser: bit-exact payload bytes and scales, head-wise and token-wise, bf16 and fp16, both numerics;
 *   - attention: against the reference's eager op output O1 to within its own bf16 rounding.
 * The reference's CUDA/ThunderKittens kernel itself (src/quantum_attn/tk/attention.py:97-349) cannot be
 * built here (nvcc + un-vendored submodule) -- for that kernel parity is pinned only through O1.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define QO_FMT_E4M3 0
#define QO_FMT_E5M2 1
#define QO_FMT_BF16 2
#define QO_FMT_FP16 3

/* ----------------------------------------------------------------------------------------------- */
/* scalar codecs                                                                                    */
/* ----------------------------------------------------------------------------------------------- */
static inline float bits_to_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f32_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

float qo_bf16_to_f32(uint16_t b) { return bits_to_f32((uint32_t)b << 16); }

uint16_t qo_f32_to_bf16(float f) { /* RNE, NaN preserved */
    uint32_t u = f32_to_bits(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

float qo_fp16_to_f32(uint16_t h) {
    uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31, m = h & 1023;
    if (e == 31) return bits_to_f32(s | 0x7f800000u | (m << 13));
    if (e == 0) { float v = ldexpf((float)m, -24); return s ? -v : v; }
    return bits_to_f32(s | ((e + 112) << 23) | (m << 13));
}

/* generic RNE rounding of a finite float to a binary format with `mbits` mantissa bits, min normal
 * exponent `emin` (value 2^emin) -- returns the rounded value as a float (exact). */
static float round_to_format(float x, int mbits, int emin) {
    if (x == 0.0f || isnan(x) || isinf(x)) return x;
    int e;
    (void)frexpf(fabsf(x), &e); /* |x| = f * 2^e, f in [0.5,1) -> exponent of leading bit = e-1 */
    int lead = e - 1;
    if (lead < emin) lead = emin;       /* subnormal range: fixed quantum */
    float quantum = ldexpf(1.0f, lead - mbits);
    float q = x / quantum;              /* exact (power of two) */
    float r = nearbyintf(q);            /* default rounding mode = RNE */
    return r * quantum;
}

uint16_t qo_f32_to_fp16(float f) { /* RNE with overflow to inf */
    uint32_t s = (f32_to_bits(f) >> 16) & 0x8000u;
    if (isnan(f)) return (uint16_t)(s | 0x7e00u);
    float r = round_to_format(f, 10, -14);
    float a = fabsf(r);
    if (a > 65504.0f) return (uint16_t)(s | 0x7c00u);
    if (a == 0.0f) return (uint16_t)s;
    int e; float fr = frexpf(a, &e); /* a = fr*2^e */
    int lead = e - 1;
    if (lead < -14) { uint32_t m = (uint32_t)ldexpf(a, 24); return (uint16_t)(s | m); }
    uint32_t m = (uint32_t)ldexpf(fr * 2.0f - 1.0f, 10);
    return (uint16_t)(s | ((uint32_t)(lead + 15) << 10) | m);
}

float qo_fp8_to_f32(uint8_t b, int fmt) {
    int s = b >> 7;
    float v;
    if (fmt == QO_FMT_E4M3) {
        int e = (b >> 3) & 15, m = b & 7;
        if (e == 15 && m == 7) v = NAN;
        else if (e == 0) v = ldexpf((float)m, -9);
        else v = ldexpf(1.0f + (float)m / 8.0f, e - 7);
    } else {
        int e = (b >> 2) & 31, m = b & 3;
        if (e == 31) v = m ? NAN : INFINITY;
        else if (e == 0) v = ldexpf((float)m, -16);
        else v = ldexpf(1.0f + (float)m / 4.0f, e - 15);
    }
    return s ? -v : v;
}

/* float -> fp8, RNE; out-of-range behaviour of torch's cast: e4m3fn -> NaN (0x7f), e5m2 -> inf. */
uint8_t qo_f32_to_fp8(float x, int fmt) {
    uint8_t s = (uint8_t)((f32_to_bits(x) >> 24) & 0x80u);
    if (fmt == QO_FMT_E4M3) {
        if (isnan(x)) return (uint8_t)(s | 0x7f);
        float r = fabsf(round_to_format(x, 3, -6));
        if (r > 448.0f) return (uint8_t)(s | 0x7f);
        if (r == 0.0f) return s;
        int e; float fr = frexpf(r, &e); int lead = e - 1;
        if (lead < -6) return (uint8_t)(s | (uint8_t)ldexpf(r, 9));
        return (uint8_t)(s | ((lead + 7) << 3) | (uint8_t)ldexpf(fr * 2.0f - 1.0f, 3));
    } else {
        if (isnan(x)) return (uint8_t)(s | 0x7f);
        float r = fabsf(round_to_format(x, 2, -14));
        if (r > 57344.0f) return (uint8_t)(s | 0x7c);
        if (r == 0.0f) return s;
        int e; float fr = frexpf(r, &e); int lead = e - 1;
        if (lead < -14) return (uint8_t)(s | (uint8_t)ldexpf(r, 16));
        return (uint8_t)(s | ((lead + 15) << 2) | (uint8_t)ldexpf(fr * 2.0f - 1.0f, 2));
    }
}

static inline float load16(const uint16_t* p, long i, int dt) {
    return dt == QO_FMT_BF16 ? qo_bf16_to_f32(p[i]) : qo_fp16_to_f32(p[i]);
}
static inline float round16(float x, int dt) {
    return dt == QO_FMT_BF16 ? qo_bf16_to_f32(qo_f32_to_bf16(x)) : qo_fp16_to_f32(qo_f32_to_fp16(x));
}
static inline float fp8_max(int fmt) { return fmt == QO_FMT_E4M3 ? 448.0f : 57344.0f; }

/* ----------------------------------------------------------------------------------------------- */
/* quantiser -- nn.py:14-19                                                                          */
/*   scale = t.abs().amax(dims, keepdim).mul(1/q_max).clamp_min(eps_f32)                             */
/*   t8    = (t / scale).clamp(-q_max, q_max).to(fp8);   returns (t8, scale.squeeze().float())       */
/* `groups` scale groups of `inner` contiguous elements (head-wise: groups=B*H, inner=S*D;           */
/*  token-wise: groups=B*H*S, inner=D).                                                              */
/* numerics 0 = "compiled" (Inductor: fp32 amax*(1/q_max), quotient fp32 then rounded to the input   */
/*              dtype before the clamp/cast) ; 1 = "eager" (every op rounds to the input dtype).      */
/* ----------------------------------------------------------------------------------------------- */
int qo_quantize_fp8(const uint16_t* x, int in_dtype, long groups, long inner, int fmt, int numerics,
                    uint8_t* out8, float* scale_out) {
    if ((in_dtype != QO_FMT_BF16 && in_dtype != QO_FMT_FP16) || (fmt != QO_FMT_E4M3 && fmt != QO_FMT_E5M2)) return -1;
    const float qmax = fp8_max(fmt);
    const float inv_qmax = (float)(1.0 / (double)qmax);
    const float eps = 1.1920928955078125e-07f; /* torch.finfo(torch.float32).eps */
#pragma omp parallel for schedule(static)
    for (long g = 0; g < groups; g++) {
        const uint16_t* xg = x + g * inner;
        float amax = 0.0f;
        int has_nan = 0;
        for (long i = 0; i < inner; i++) {
            float a = fabsf(load16(xg, i, in_dtype));
            if (isnan(a)) has_nan = 1;
            if (a > amax) amax = a;
        }
        if (has_nan) amax = NAN;
        float scale;
        if (numerics == 0) {
            scale = amax * inv_qmax;
            if (!(scale >= eps)) scale = isnan(scale) ? scale : eps; /* clamp_min propagates NaN */
        } else {
            scale = round16(amax * inv_qmax, in_dtype);
            if (!(scale >= eps)) scale = isnan(scale) ? scale : round16(eps, in_dtype);
        }
        scale_out[g] = scale;
        for (long i = 0; i < inner; i++) {
            float t = load16(xg, i, in_dtype);
            float q = round16(t / scale, in_dtype);
            if (q > qmax) q = qmax;
            if (q < -qmax) q = -qmax;
            out8[g * inner + i] = qo_f32_to_fp8(q, fmt);
        }
    }
    return 0;
}

/* ----------------------------------------------------------------------------------------------- */
/* attention forward -- ops.py:64-95 / ops.py:17-29 evaluated in fp64                                */
/*   q,k: [B,Hq|Hkv,S,D] in q_fmt/k_fmt (fp8 byte or 16-bit); v: [B,Hkv,Skv,D] in v_fmt.             */
/*   scale_q/scale_k: NULL, or fp32 [B,H] (scale_mode 0, head-wise) / [B,H,S] (scale_mode 1).        */
/*   scale_v: NULL or fp32 [B,Hkv] (build extension: quantised V).                                   */
/*   sm_scale <= 0 -> 1/sqrt(D) (aten default).  causal = 1: keep key j <= query i (aten top-left);  */
/*   causal = 1 + r0: the Sq rows are rows r0 .. r0+Sq-1 of a longer causal problem (key j <= r0 + i) */
/*   -- lets a test check a band of rows of a long sequence without attending the rows before it.     */
/*   out: fp32 [B,Hq,Sq,D]; lse (optional): fp32 [B,Hq,Sq] natural-log-sum-exp of the scaled scores.  */
/* ----------------------------------------------------------------------------------------------- */
static void dequant_rows(const void* src, int fmt, long n, double mul, float* dst) {
    if (fmt == QO_FMT_E4M3 || fmt == QO_FMT_E5M2) {
        const uint8_t* p = (const uint8_t*)src;
        for (long i = 0; i < n; i++) dst[i] = (float)((double)qo_fp8_to_f32(p[i], fmt) * mul);
    } else {
        const uint16_t* p = (const uint16_t*)src;
        for (long i = 0; i < n; i++) dst[i] = (float)((double)load16(p, i, fmt) * mul);
    }
}
static long elt_size(int fmt) { return (fmt == QO_FMT_E4M3 || fmt == QO_FMT_E5M2) ? 1 : 2; }

int qo_attention_forward(const void* q, const void* k, const void* v, int q_fmt, int k_fmt, int v_fmt,
                         const float* scale_q, const float* scale_k, const float* scale_v, int scale_mode,
                         int B, int Hq, int Hkv, int Sq, int Skv, int D, int causal, float sm_scale,
                         float* out, float* lse) {
    if (Hkv <= 0 || Hq % Hkv != 0) return -1;
    const int hr = Hq / Hkv;
    const double sm = sm_scale > 0.0f ? (double)sm_scale : 1.0 / sqrt((double)D);
    int err = 0;
#pragma omp parallel
    {
        float* kf = (float*)malloc(sizeof(float) * (size_t)Skv * D);
        float* vf = (float*)malloc(sizeof(float) * (size_t)Skv * D);
        float* qf = (float*)malloc(sizeof(float) * (size_t)D);
        double* s = (double*)malloc(sizeof(double) * (size_t)Skv);
        double* acc = (double*)malloc(sizeof(double) * (size_t)D);
        if (!kf || !vf || !qf || !s || !acc) err = -2;
#pragma omp for schedule(dynamic, 1) collapse(2)
        for (int b = 0; b < B; b++) {
            for (int h = 0; h < Hq; h++) {
                if (err) continue;
                const int hk = h / hr;
                const long kv_off = ((long)b * Hkv + hk) * Skv * D;
                /* de-quantise K and V of this head once; scales applied in fp64 then kept as float */
                if (scale_k && scale_mode == 1) {
                    for (int j = 0; j < Skv; j++)
                        dequant_rows((const char*)k + (kv_off + (long)j * D) * elt_size(k_fmt), k_fmt, D,
                                     (double)scale_k[((long)b * Hkv + hk) * Skv + j], kf + (long)j * D);
                } else {
                    dequant_rows((const char*)k + kv_off * elt_size(k_fmt), k_fmt, (long)Skv * D,
                                 scale_k ? (double)scale_k[(long)b * Hkv + hk] : 1.0, kf);
                }
                dequant_rows((const char*)v + kv_off * elt_size(v_fmt), v_fmt, (long)Skv * D,
                             scale_v ? (double)scale_v[(long)b * Hkv + hk] : 1.0, vf);
                for (int i = 0; i < Sq; i++) {
                    const long q_off = (((long)b * Hq + h) * Sq + i) * D;
                    double sq = 1.0;
                    if (scale_q) sq = scale_mode == 1 ? (double)scale_q[((long)b * Hq + h) * Sq + i] : (double)scale_q[(long)b * Hq + h];
                    dequant_rows((const char*)q + q_off * elt_size(q_fmt), q_fmt, D, 1.0, qf);
                    const int jmax = causal > 0 ? ((long)i + causal < (long)Skv ? i + causal : Skv) : Skv;
                    double m = -INFINITY;
                    for (int j = 0; j < jmax; j++) {
                        const float* kr = kf + (long)j * D;
                        double d = 0.0;
                        for (int c = 0; c < D; c++) d += (double)qf[c] * (double)kr[c];
                        d *= sq * sm;
                        s[j] = d;
                        if (d > m) m = d;
                    }
                    double l = 0.0;
                    for (int c = 0; c < D; c++) acc[c] = 0.0;
                    for (int j = 0; j < jmax; j++) {
                        double p = exp(s[j] - m);
                        l += p;
                        const float* vr = vf + (long)j * D;
                        for (int c = 0; c < D; c++) acc[c] += p * (double)vr[c];
                    }
                    for (int c = 0; c < D; c++) out[q_off + c] = jmax > 0 ? (float)(acc[c] / l) : NAN;
                    if (lse) lse[((long)b * Hq + h) * Sq + i] = jmax > 0 ? (float)(m + log(l)) : -INFINITY;
                }
            }
        }
        free(kf); free(vf); free(qf); free(s); free(acc);
    }
    return err;
}

/* de-quantise a whole fp8 tensor to fp32 (helper for tests) */
void qo_fp8_to_f32_array(const uint8_t* src, long n, int fmt, float* dst) {
    for (long i = 0; i < n; i++) dst[i] = qo_fp8_to_f32(src[i], fmt);
}
void qo_f32_to_fp8_array(const float* src, long n, int fmt, uint8_t* dst) {
    for (long i = 0; i < n; i++) dst[i] = qo_f32_to_fp8(src[i], fmt);
}
void qo_f32_to_bf16_array(const float* src, long n, uint16_t* dst) {
    for (long i = 0; i < n; i++) dst[i] = qo_f32_to_bf16(src[i]);
}
int qo_abi_version(void) { return 1; }
