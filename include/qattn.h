/*
 * qattn.h -- C ABI of libqattn_hip.so: MI355X (gfx950) FP8 fused attention forward + bf16/fp16->fp8 quant pre-pass: the drop-in
 * boundary for the ONE hot path of WaveSpeedAI/QuantumAttention (reference @ 2025-02-22; paths relative to src/quantum_attn/).  Replaces:
 *   qattn_fp8_attention_forward_rowmajor   pybind `attention_forward(q, k, v, scale_q, scale_k, causal)` (tk/attention.py:355-360, 688-702)
 *                                          behind op `quantum_attn::fp8_attention_forward` (ops.py:98-121): launcher + `fwd_attend_ker`
 *                                          (tk/attention.py:97-349, 355-647) in ONE call;  qattn_fp8_attention_forward: the same on
 *                                          operands already in this library's fragment layouts (K / V re-laid once, reused)
 *   qattn_quant_fp8 / _quant_qkv_fp8       `_dynamically_quantize_fp8` (nn.py:14-19; callers nn.py:410-418, 22-42)
 *   qattn_fp8_quant_attention_forward[_ex] the whole `_fp8_attention_wrapper` step (nn.py:394-430) for 16-bit inputs in one call
 *   qattn_attention_forward_16, _pack16    the non-fp8 build of the kernel behind `quantum_attn::attention_forward` (ops.py:17-45)
 * Conventions (as the reference launcher, tk/attention.py:362-465): plain C; pointers are DEVICE pointers on the current HIP device; tensors
 * dense [B,H,S,D] row-major unless a fragment layout is named; `stream` = hipStream_t as void* (NULL = default).  Calls enqueue work and
 * return 0 or a negative QATTN_ERR_* code: no host sync, no allocation, no throw, no environment variable, graph-capture safe.  (One
 * exception: a causal call on the templated kernel forks its early rows onto ONE internal stream per host thread and device and joins it
 * back with events; created on first use outside a capture, destroyed at thread exit.)
 * QATTN_LAYOUT_KFRAG / _VFRAG / _K16FRAG / _V16FRAG: K and V re-laid per 64-key chunk into the MFMA A-operand order (private to the
 * library: produced by qattn_quant_fp8 / qattn_pack_fp8 / qattn_pack16, S zero-padded to a multiple of 64; byte maps: csrc/qattn_common.h).
 */
#ifndef QATTN_H_
#define QATTN_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QATTN_ABI_VERSION 8   /* 7: lse / row_path on ..._forward_ex, ..._rowmajor, qattn_describe_path; 8: qattn_strided.h */

#define QATTN_FMT_E4M3 0 /* OCP float8_e4m3fn (torch.float8_e4m3fn) */
#define QATTN_FMT_E5M2 1 /* OCP float8_e5m2   (torch.float8_e5m2)   */
#define QATTN_FMT_BF16 2
#define QATTN_FMT_FP16 3

/* scale granularity (nn.py:410-414): one fp32 per (b,h) [B,H], or one per row [B,H,S] */
#define QATTN_SCALE_HEAD 0
#define QATTN_SCALE_TOKEN 1

#define QATTN_LAYOUT_ROWMAJOR 0
#define QATTN_LAYOUT_KFRAG 1
#define QATTN_LAYOUT_VFRAG 2
#define QATTN_LAYOUT_K16FRAG 3
#define QATTN_LAYOUT_V16FRAG 4

/* quantiser numerics (both bit-exact): the reference's compiled GPU path / its eager arithmetic */
#define QATTN_NUMERICS_COMPILED 0
#define QATTN_NUMERICS_EAGER 1

/* P in the second GEMM (the reference keeps P and V 16-bit, tk/attention.py:72,286,318; here P is e4m3).  AUTO (default): one-term P,
 * rows with a weight > 1/24 (R = l/p_max < 24) or an effective key count < 192 are recomputed with more precision (PATH TABLE `precise`);
 * FAST: no check; ACCURATE: the precise pass everywhere. */
#define QATTN_PRECISION_AUTO 0
#define QATTN_PRECISION_FAST 1
#define QATTN_PRECISION_ACCURATE 2

/* log-sum-exp output: NATURAL = dense fp32 [B,Hq,Sq], ln sum_j exp(score_j); REFERENCE = the reference's (disabled) vector,
 * tk/attention.py:333-346,439-446: -sqrt(D) * NATURAL, (b,h) rows qattn_lse_row_stride() = ceil(Sq*4/16)*16/4 floats apart. */
#define QATTN_LSE_NATURAL 0
#define QATTN_LSE_REFERENCE 1

/* `row_path` codes (..._forward_ex): the numerics that produced a row = the oracle a parity test holds it against.  ONE_TERM / TWO_TERM:
 * e4m3 P (one / hi + lo terms) on the fp8 V -> fp64 SDPA on the quantised q, k, v;  V16: 16-bit P on the caller's ORIGINAL 16-bit V. */
#define QATTN_PATH_ONE_TERM 0
#define QATTN_PATH_TWO_TERM 1
#define QATTN_PATH_V16 2

#define QATTN_OK 0
#define QATTN_ERR_INVALID_ARG (-1)     /* NULL pointer, non-positive dimension, unknown enum */
#define QATTN_ERR_UNSUPPORTED_DIM (-2) /* head_dim not in {64,128,256} (nn.py:45-49), or Hq % Hkv != 0 */
#define QATTN_ERR_UNSUPPORTED_FMT (-3) /* format / layout combination not implemented */
#define QATTN_ERR_WORKSPACE (-4)       /* workspace missing or too small */
#define QATTN_ERR_LAUNCH (-5)          /* HIP launch failed */
#define QATTN_ERR_DEVICE (-6)          /* current device is not gfx950 */

int qattn_abi_version(void);
const char* qattn_strerror(int code);
/* 0 on gfx950, else QATTN_ERR_DEVICE; replaces `cuda_capability_compare("ge", 9, 0)` (utils/checks.py:57-64, nn.py:214) */
int qattn_check_device(void);

size_t qattn_fp8_tensor_bytes(int layout, int B, int H, int S, int D);
size_t qattn_quant_workspace_bytes(int B, int H, int S, int D, int scale_mode);

/* Quant pre-pass (nn.py:14-19): scale = clamp_min(amax|x| / fmax, eps_f32); x8 = fp8(clamp(x / scale, +-fmax)).
 * x [B,H,S,D] bf16 / fp16 (in_fmt); x8 in `out_layout`, out_fmt E4M3 (reference) or E5M2; scale fp32 [B,H] or [B,H,S]. */
int qattn_quant_fp8(const void* x, int in_fmt, void* x8, float* scale, int B, int H, int S, int D, int out_fmt,
        int scale_mode, int numerics, int out_layout, void* workspace, size_t workspace_bytes,
        void* stream);

/* q, k, v in ONE abs-max + ONE quantise launch (q8 row-major, k8 KFRAG, v8 VFRAG head-wise) = three qattn_quant_fp8 calls */
size_t qattn_quant_qkv_workspace_bytes(int B, int Hq, int Hkv);
int qattn_quant_qkv_fp8(const void* q, const void* k, const void* v, int in_fmt, void* q8, void* k8, void* v8,
        float* scale_q, float* scale_k, float* scale_v, int B, int Hq, int Hkv, int Sq, int Skv, int D,
        int out_fmt, int scale_mode, int numerics, void* workspace, size_t workspace_bytes, void* stream);

/* row-major fp8 [B,H,S,D] -> KFRAG / VFRAG (byte permutation) */
int qattn_pack_fp8(const void* x8_rowmajor, void* x8_packed, int B, int H, int S, int D, int out_layout, void* stream);

/*
 * FP8 fused attention forward:  O = softmax(sm_scale (sq Q8)(sk K8)^T [+ causal mask]) (sv V8), flash-style, fp32 accumulation / softmax.
 *   q8 [B,Hq,Sq,D] fp8 row-major; k8 [B,Hkv,Skv,D] KFRAG; v8 VFRAG (v_fmt = qk_fmt: both GEMMs on FP8 MFMA) -- or, with v_fmt = BF16 / FP16
 *   = out_fmt, the ORIGINAL 16-bit V row-major, scale_v NULL: every row runs the reference's P.V numerics (16-bit P and V; ~1.5x the time).
 *   out [B,Hq,Sq,D] bf16 / fp16; lse NULL or the per-row vector; scale_q fp32 [B,Hq] / [B,Hq,Sq], scale_k likewise, scale_v [B,Hkv] or NULL;
 *   sm_scale <= 0 = 1/sqrt(D) (tk/attention.py:208-210); is_causal: key j <= query i (top-left); workspace: needed for AUTO, may be NULL
 *   otherwise (static block hand-out: same bits, a few per cent slower on long sequences).  Numerics: PATH TABLE, separate / separate16.
 */
size_t qattn_attention_workspace_bytes(int B, int Hq, int Sq);
size_t qattn_lse_row_stride(int Sq, int lse_layout);
int qattn_fp8_attention_forward(const void* q8, const void* k8, const void* v8, void* out, float* lse,
        const float* scale_q, const float* scale_k, const float* scale_v, int B, int Hq,
        int Hkv, int Sq, int Skv, int D, int qk_fmt, int v_fmt, int out_fmt, int scale_mode,
        int is_causal, float sm_scale, int precision, int lse_layout, void* workspace,
        size_t workspace_bytes, void* stream);

/*
 * The pybind function's contract in ONE call -- `attention_forward(q, k, v, scale_q, scale_k, causal)` (tk/attention.py:357-360, 419-437):
 * q8 / k8 fp8 ROW-MAJOR, v16 bf16 / fp16 (= the output's format).  The K re-lay and the V handling happen inside, in `workspace`:
 * pv_fmt = qk_fmt quantises V head-wise (both GEMMs FP8, the default); pv_fmt = v16_fmt keeps V and P 16-bit.  Bit-identical to
 * qattn_pack_fp8 + qattn_quant_fp8 + qattn_fp8_attention_forward; the Python op `fp8_attention_forward` is this call.
 */
size_t qattn_fp8_attention_rowmajor_workspace_bytes(int B, int Hq, int Hkv, int Sq, int Skv, int D);
int qattn_fp8_attention_forward_rowmajor(const void* q8, const void* k8, const void* v16, void* out, float* lse, const float* scale_q,
        const float* scale_k, int B, int Hq, int Hkv, int Sq, int Skv, int D, int qk_fmt, int v16_fmt,
        int pv_fmt, int scale_mode, int is_causal, float sm_scale, int precision, int lse_layout,
        void* workspace, size_t workspace_bytes, void* stream);

/*
 * The whole step of `_fp8_attention_wrapper` for 16-bit inputs (nn.py:394-430) in one call: pre-pass + attention.  q8 / k8 / v8 / scale_*:
 * caller-provided outputs + scratch (qattn_quant_qkv_fp8's sizes; q8 untouched where the kernel quantises Q; scale_v = 1 where V is
 * block-scaled).  Rows that need precision run on the caller's 16-bit V: NOT the separate calls' results there (PATH TABLE).
 * qattn_vblock_exponent: exponent of a block-scaled V chunk's scale from the fp32 bits of its abs-max (E8M0 byte = e + 127).
 */
size_t qattn_fp8_quant_attention_workspace_bytes(int B, int Hq, int Hkv, int Sq);
int qattn_vblock_exponent(unsigned amax_bits, int out_fmt);
int qattn_fp8_quant_attention_forward(const void* q, const void* k, const void* v, int in_fmt, void* out, void* q8, void* k8,
        void* v8, float* scale_q, float* scale_k, float* scale_v, int B, int Hq, int Hkv, int Sq,
        int Skv, int D, int fp8_fmt, int scale_mode, int numerics, int is_causal, float sm_scale,
        int precision, void* workspace, size_t workspace_bytes, void* stream);

/*
 * The same step with a producer's per-head figures (head-wise scales only) and the optional outputs.
 *   amax_q/_k/_v  NULL or fp32 [B,H]: exact max |x| per head (what Inductor's fusion of the quantiser into the producer gives the reference,
 *       nn.py:410-418); a supplied tensor skips the abs-max launch.  Same bits as the plain call -- under AUTO when ssq_q / ssq_k (fp32 [B,H]
 *       sums of x^2, both or neither) come along; without them wide heads start one-term (same bound).  Too large is safe, too small clips.
 *   lse / lse_layout  NULL or the per-row log-sum-exp written BY THE SAME LAUNCH as `out` (reference: l_vec, tk/attention.py:79,85,333-346,
 *       439-452).  D = 128 head-wise FP8 sweep: from the sum of the e4m3-rounded weights the second GEMM consumed (mean offset removed), `out`
 *       unchanged, within 2e-2 (natural log); 16-bit-V rows (sums of the rounded P) 4e-3, others 2e-3.
 *   row_path  NULL or uint8 [B,Hq,Sq] (QATTN_PATH_*): test / debug output; no cost when NULL, `out` does not depend on it.
 */
int qattn_fp8_quant_attention_forward_ex(const void* q, const void* k, const void* v, int in_fmt, void* out, void* q8, void* k8,
        void* v8, float* scale_q, float* scale_k, float* scale_v, const float* amax_q,
        const float* amax_k, const float* amax_v, const float* ssq_q, const float* ssq_k, int B,
        int Hq, int Hkv, int Sq, int Skv, int D, int fp8_fmt, int scale_mode, int numerics,
        int is_causal, float sm_scale, int precision, float* lse, int lse_layout,
        unsigned char* row_path, void* workspace, size_t workspace_bytes, void* stream);

/*
 * PATH TABLE -- which numerics each entry runs.  qattn_describe_path() (host-only) answers from the predicates the dispatch itself uses;
 * tests/test_cpu_boundary.py::test_path_table_matches_dispatch walks these rows against it.  bf16 / fp16, e4m3 / e5m2 change no row.
 *
 * | entry      | D      | scales | Skv      | kernel | q_quant | v_format | sweep_p | precise  | early      | start   | lse       |
 * |------------|--------|--------|----------|--------|---------|----------|---------|----------|------------|---------|-----------|
 * | fused      | 128    | head   | <=16384  | v2     | kernel  | block    | byte    | v16      | v16-inline | moments | quantised |
 * | fused      | 128    | head   | >16384   | v2     | kernel  | head     | byte    | v16      | v16-inline | moments | quantised |
 * | fused      | 64,256 | head   | <=16384  | v4     | prepass | block    | byte    | two-term | v16-launch | moments | exact*    |
 * | fused      | 64,256 | head   | >16384   | v4     | prepass | head     | byte    | two-term | v16-launch | moments | exact*    |
 * | fused      | any    | token  | any      | v4     | prepass | head     | byte    | two-term | v16-launch | keys    | exact*    |
 * | separate   | 128    | head   | any      | v2     | caller  | head     | byte    | two-term | two-term   | keys    | exact*    |
 * | separate   | 64,256 | head   | any      | v4     | caller  | head     | byte    | two-term | two-term   | keys    | exact*    |
 * | separate   | any    | token  | any      | v4     | caller  | head     | byte    | two-term | two-term   | keys    | exact*    |
 * | separate16 | any    | any    | any      | pv16   | caller  | 16bit    | p16     | none     | none       | none    | exact     |
 *
 * entry: fused = ..._quant_attention_forward[_ex]; separate / separate16 = ..._attention_forward / _rowmajor with an fp8 / a 16-bit V.
 * kernel: v2 = csrc/qattn_attn_v2.hip (hand-scheduled), v4 = qattn_attn_v4.hip (templated), pv16 = qattn_pv16.h.  q_quant kernel = in the
 * attention prologue (q8 never written).  v_format block = one power-of-two scale per 64-key chunk (oracle.quantize_v_block), head = one
 * fp32 scale per head.  precise = where AUTO's flagged rows / blocks and ACCURATE go: v16 = 16-bit P on the caller's 16-bit V (rescued
 * rows with 8 <= R < 24: two-term fp8 P, QATTN_PATH_TWO_TERM); two-term = hi + lo e4m3 P on the fp8 V.  early = query blocks whose first
 * row sees < 1024 keys.  start = what picks a block's starting mode under AUTO (keys: key-count rule + first-chunk forecast; moments: + the
 * per-head sums of squares, predicted score variance >= 1.5 starts precise).  lse exact* = an LSE request switches the sweep to exact
 * exponentials (same bound, other bits); quantised = sums of the e4m3 weights, output unchanged.  AUTO's 2^-6 assumes values within ~4.5
 * of the output on the keys a one-term row keeps (weights < 1/24): it scales with the LARGEST value -- heavy-tailed V belongs on ACCURATE.
 */
enum { QATTN_ENTRY_SEPARATE, QATTN_ENTRY_SEPARATE_V16, QATTN_ENTRY_FUSED };           /* `entry` argument */
enum { QATTN_KERNEL_V2, QATTN_KERNEL_V4, QATTN_KERNEL_PV16 };                           /* the table's columns, values in the table's words */
enum { QATTN_QQUANT_CALLER, QATTN_QQUANT_PREPASS, QATTN_QQUANT_KERNEL };
enum { QATTN_VFORMAT_HEAD, QATTN_VFORMAT_BLOCK, QATTN_VFORMAT_16BIT };
enum { QATTN_SWEEP_BYTE, QATTN_SWEEP_EXACT, QATTN_SWEEP_P16 };
enum { QATTN_PRECISE_TWO_TERM, QATTN_PRECISE_V16, QATTN_PRECISE_NONE };
enum { QATTN_EARLY_TWO_TERM, QATTN_EARLY_V16_INLINE, QATTN_EARLY_V16_LAUNCH, QATTN_EARLY_NONE };
enum { QATTN_START_KEYS, QATTN_START_MOMENTS, QATTN_START_NONE };
enum { QATTN_LSE_SRC_EXACT, QATTN_LSE_SRC_QUANTISED };
typedef struct qattn_path_desc {
    int kernel, q_quant, v_format, sweep_p, precise, early, start_mode, lse;
} qattn_path_desc;
/* 0 or a negative code; no device call.  in_fmt: QATTN_FMT_BF16 / _FP16; want_lse: the call passes lse != NULL. */
int qattn_describe_path(int entry, int D, int in_fmt, int scale_mode, int Skv, int want_lse, qattn_path_desc* desc);

/* 16-bit sibling path: the non-fp8 build of the same kernel (tk/attention.py:212,238-240,289-313) behind `quantum_attn::attention_forward`
 * (ops.py:17-45).  q / out row-major bf16 or fp16 (`fmt`); k16 / v16 re-laid by qattn_pack16 (K16FRAG / V16FRAG); both GEMMs on
 * v_mfma_f32_32x32x16_{bf16,f16}.  fast_exp = 1: linear-mantissa 2^x (1.8 % rms per weight) for rows known to be flat.  lse: NULL or [B,Hq,Sq]. */
size_t qattn_16bit_tensor_bytes(int layout, int B, int H, int S, int D);
int qattn_pack16(const void* x_rowmajor, void* x_packed, int B, int H, int S, int D, int out_layout, void* stream);
int qattn_attention_forward_16(const void* q, const void* k16, const void* v16, void* out, float* lse, int B, int Hq,
        int Hkv, int Sq, int Skv, int D, int fmt, int is_causal, float sm_scale, int fast_exp,
        void* stream);

/* strided views of q, k, v: qattn_strided.h; measurement aids (clock stamps, launch timing, probe): qattn_measure.h */

#ifdef __cplusplus
}
#endif
#endif /* QATTN_H_ */
