/*
 * qattn.h -- C ABI of libqattn_hip.so: MI355X (gfx950) FP8 fused attention forward + bf16/fp16->fp8 quant pre-pass.
 * The drop-in boundary for the ONE hot path of WaveSpeedAI/QuantumAttention (reference @ 2025-02-22; paths below are relative to
 * src/quantum_attn/).  What each entry replaces:
 *   qattn_fp8_attention_forward_rowmajor   the pybind entry `attention_forward(q, k, v, scale_q, scale_k, causal)` (tk/attention.py:355-360,
 *                                          688-702) behind the op `quantum_attn::fp8_attention_forward` (ops.py:98-121), i.e. the launcher +
 *                                          kernel `fwd_attend_ker<D,causal,..>` (tk/attention.py:97-349, 355-647) -- ONE call, row-major tensors
 *   qattn_fp8_attention_forward            the same on operands already in this library's fragment layouts (K / V re-laid once, reused)
 *   qattn_quant_fp8 / _quant_qkv_fp8       `_dynamically_quantize_fp8` (nn.py:14-19) as invoked by `_fp8_attention_wrapper` (nn.py:410-418) /
 *                                          `dynamically_quantize_fp8` (nn.py:22-42) -- Inductor-generated Triton in the reference
 *   qattn_fp8_quant_attention_forward[_ex] the whole `_fp8_attention_wrapper` step (nn.py:394-430) for 16-bit inputs in one call
 *   qattn_attention_forward_16, _pack16    the non-fp8 build of the kernel behind `quantum_attn::attention_forward` (ops.py:17-45)
 *   qattn_pack_fp8, _describe_path, _profile_*, _mfma_probe, _*_stamped   no reference counterpart (layout helper, query, measurement aids)
 *
 * Conventions (as the reference launcher, tk/attention.py:362-465, unless noted): plain C, no torch types; every pointer is a DEVICE
 * pointer on the current HIP device; tensors are dense [B, H, S, D] row-major unless a fragment layout is named; `stream` is a
 * hipStream_t passed as void* (NULL = default stream).  Calls enqueue work and return: no host synchronisation, no allocation,
 * graph-capture safe, no environment variable read.  One exception to "no allocation": a causal call on the templated kernel runs its
 * early rows on ONE internal non-blocking stream per host thread and device (forked from / joined to `stream` with events; created on
 * the thread's first such call outside a capture, destroyed when the thread exits; if creation fails everything stays on `stream`).
 * Every function returns 0 or a negative QATTN_ERR_* code (qattn_strerror names it); nothing is thrown.
 *
 * Fragment layouts (private; produced by qattn_quant_fp8 / qattn_pack_fp8; S zero-padded to Sp = 64*ceil(S/64); D*Sp bytes per (b,h)):
 *   QATTN_LAYOUT_KFRAG  chunk of 64 keys = [t:2][s:D/64][hh:2][half:2][key:32][16 B]; byte j of a piece = K[64c + 32t + key][64s + 32hh + 16half + j]
 *   QATTN_LAYOUT_VFRAG  chunk = [m:D/32][hh:2][half:2][d:32][16 B]; byte 4w+i of a piece = V[64c + 32half + 8w + 4hh + i][32m + d]
 *   = the A operands of v_mfma_f32_32x32x64_f8f6f4 for S^T = K.Q^T and O^T = V^T.P^T: linear for LDS-DMA, conflict-free for ds_read_b128.
 *   QATTN_LAYOUT_K16FRAG chunk = [t:2][s:D/16][hh:2][key:32][8 elts], piece = K[64c + 32t + key][16s + 8hh + (0..7)]          (16-bit path)
 *   QATTN_LAYOUT_V16FRAG chunk = [t:2][m:D/32][s:2][hh:2][d:32][8 elts], elt j = V[64c + 32t + 16s + 8(j>>2) + 4hh + (j&3)][32m + d]
 */
#ifndef QATTN_H_
#define QATTN_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QATTN_ABI_VERSION 7   /* 7 (round 6): lse / lse_layout / row_path on ..._forward_ex, ..._forward_rowmajor, qattn_describe_path */

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

/* quantiser numerics: what the reference's compiled GPU path computes (fp32 scale, quotient rounded to the
 * input dtype) / the reference's eager arithmetic (everything in the input dtype).  Both bit-exact. */
#define QATTN_NUMERICS_COMPILED 0
#define QATTN_NUMERICS_EAGER 1

/* How the softmax weights P enter the second fp8 GEMM.  The reference keeps P and V 16-bit (tk/attention.py:72,286,318); here P is e4m3,
 * and ONE e4m3 term is accurate enough only for rows whose weight is spread over many keys:
 *   AUTO (default)  one-term P, checked per row: rows whose largest weight exceeds 1/24 (R = l / p_max < 24) or whose effective key
 *                   count is below 192 are recomputed with more precision (a few rows: gathered and rescued; many: the 256-row block).
 *   FAST            one-term P wherever a row sees >= 1024 keys; no check.       ACCURATE  the precise pass everywhere.
 * "More precision" and what early rows run: the `precise` / `early` columns of the PATH TABLE below. */
#define QATTN_PRECISION_AUTO 0
#define QATTN_PRECISION_FAST 1
#define QATTN_PRECISION_ACCURATE 2

/* optional log-sum-exp output: NATURAL = dense fp32 [B,Hq,Sq], ln sum_j exp(score_j); REFERENCE = the vector the reference defines in
 * its (disabled) epilogue, tk/attention.py:333-346,439-446: L = -sqrt(D) * NATURAL, rows of consecutive (b,h) spaced
 * qattn_lse_row_stride(Sq, REFERENCE) = ceil(Sq*4/16)*16/4 floats (the reference hard-codes -8 / -11.3137; here -sqrt(D)). */
#define QATTN_LSE_NATURAL 0
#define QATTN_LSE_REFERENCE 1

/* codes of the optional `row_path` output (qattn_fp8_quant_attention_forward_ex): the numerics that produced the row -- i.e. the oracle a
 * parity test must hold it against.  ONE_TERM / TWO_TERM: e4m3 P (one / hi + lo terms) on the fp8 V -> fp64 SDPA on the quantised q, k, v;
 * V16: 16-bit P on the caller's ORIGINAL 16-bit V (tk/attention.py:72,286,318) -> fp64 SDPA on the quantised q, k and the 16-bit v. */
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
/* 0 on gfx950 (MI355X), else QATTN_ERR_DEVICE; replaces `cuda_capability_compare("ge", 9, 0)` (utils/checks.py:57-64, nn.py:214). */
int qattn_check_device(void);

/* bytes of an fp8 tensor [B,H,S,D] in `layout`; scratch bytes of qattn_quant_fp8 (0 for token-wise). */
size_t qattn_fp8_tensor_bytes(int layout, int B, int H, int S, int D);
size_t qattn_quant_workspace_bytes(int B, int H, int S, int D, int scale_mode);

/* Quant pre-pass (nn.py:14-19): scale = clamp_min(amax|x| / fmax, eps_f32); x8 = fp8(clamp(x / scale, +-fmax)).
 * x [B,H,S,D] bf16 / fp16 (in_fmt); x8 in `out_layout`, out_fmt E4M3 (reference) or E5M2; scale fp32 [B,H] or [B,H,S]. */
int qattn_quant_fp8(const void* x, int in_fmt, void* x8, float* scale, int B, int H, int S, int D, int out_fmt,
                    int scale_mode, int numerics, int out_layout, void* workspace, size_t workspace_bytes,
                    void* stream);

/* q, k and v of one attention call in ONE abs-max + ONE quantise launch: q8 row-major, k8 KFRAG, v8 VFRAG (v always head-wise); the same
 * numerics as three qattn_quant_fp8 calls.  What `_fp8_attention_wrapper` does at nn.py:410-418, plus the build's quantised V. */
size_t qattn_quant_qkv_workspace_bytes(int B, int Hq, int Hkv);
int qattn_quant_qkv_fp8(const void* q, const void* k, const void* v, int in_fmt, void* q8, void* k8, void* v8,
                        float* scale_q, float* scale_k, float* scale_v, int B, int Hq, int Hkv, int Sq, int Skv, int D,
                        int out_fmt, int scale_mode, int numerics, void* workspace, size_t workspace_bytes, void* stream);

/* Re-lay a row-major fp8 tensor [B,H,S,D] into QATTN_LAYOUT_KFRAG / _VFRAG (byte permutation). */
int qattn_pack_fp8(const void* x8_rowmajor, void* x8_packed, int B, int H, int S, int D, int out_layout, void* stream);

/*
 * FP8 fused attention forward:  O = softmax(sm_scale (sq Q8)(sk K8)^T [+ causal mask]) (sv V8), flash-style.
 *   q8 [B,Hq,Sq,D] fp8 row-major; k8 [B,Hkv,Skv,D] KFRAG; v8 VFRAG (v_fmt = qk_fmt: both GEMMs on FP8 MFMA) -- or, with
 *   v_fmt = QATTN_FMT_BF16 / _FP16 = out_fmt, the ORIGINAL 16-bit V, dense row-major, scale_v NULL: every row then runs the reference
 *   kernel's own P.V numerics (fp8 QK^T, 16-bit P, 16-bit V; about 1.5x the time).
 *   out [B,Hq,Sq,D] bf16 / fp16, written in full; lse NULL or the per-row vector in `lse_layout`; scale_q fp32 [B,Hq] / [B,Hq,Sq],
 *   scale_k likewise, scale_v fp32 [B,Hkv] or NULL (= 1); sm_scale <= 0 selects 1/sqrt(D) (tk/attention.py:208-210); is_causal: key j
 *   <= query i (top-left; the reference requires Sq == Skv); workspace: qattn_attention_workspace_bytes() bytes -- needed for
 *   QATTN_PRECISION_AUTO, may be NULL otherwise (static block hand-out then: a few per cent slower on long sequences, same bits).
 * Accumulation, running max / sum and the softmax are fp32.  Which kernel and numerics: PATH TABLE, entries separate / separate16.
 */
size_t qattn_attention_workspace_bytes(int B, int Hq, int Sq);
size_t qattn_lse_row_stride(int Sq, int lse_layout);
int qattn_fp8_attention_forward(const void* q8, const void* k8, const void* v8, void* out, float* lse,
                                const float* scale_q, const float* scale_k, const float* scale_v, int B, int Hq,
                                int Hkv, int Sq, int Skv, int D, int qk_fmt, int v_fmt, int out_fmt, int scale_mode,
                                int is_causal, float sm_scale, int precision, int lse_layout, void* workspace,
                                size_t workspace_bytes, void* stream);

/*
 * The pybind function's contract in ONE call -- `attention_forward(q, k, v, scale_q, scale_k, causal)` on row-major tensors
 * (tk/attention.py:357-360, 419-437): q8 / k8 fp8 ROW-MAJOR, v16 bf16 / fp16 (= the output's format) row-major.  The K re-lay and the V
 * handling happen inside, in `workspace` (nothing in it outlives the call): pv_fmt = qk_fmt quantises V head-wise (both GEMMs on FP8
 * MFMA, this library's default); pv_fmt = v16_fmt keeps V and P 16-bit.  Bit-identical to qattn_pack_fp8 + qattn_quant_fp8 +
 * qattn_fp8_attention_forward; this is what the Python op `fp8_attention_forward` calls.
 */
size_t qattn_fp8_attention_rowmajor_workspace_bytes(int B, int Hq, int Hkv, int Sq, int Skv, int D);
int qattn_fp8_attention_forward_rowmajor(const void* q8, const void* k8, const void* v16, void* out, float* lse, const float* scale_q,
                                         const float* scale_k, int B, int Hq, int Hkv, int Sq, int Skv, int D, int qk_fmt, int v16_fmt,
                                         int pv_fmt, int scale_mode, int is_causal, float sm_scale, int precision, int lse_layout,
                                         void* workspace, size_t workspace_bytes, void* stream);

/*
 * The whole step of `_fp8_attention_wrapper` for 16-bit inputs (nn.py:394-430) in one call: pre-pass + attention on `stream`.
 * q8 / k8 / v8 / scale_* are caller-provided outputs + scratch with qattn_quant_qkv_fp8's sizes (q8 stays untouched where the kernel
 * quantises Q itself; scale_v is 1.0 and the chunk scales live in the workspace where V is block-scaled).  Having the caller's 16-bit V
 * at hand, the step runs the rows that need precision on it -- its results are NOT those of the separate calls there (PATH TABLE).
 * qattn_vblock_exponent: the exponent e of a block-scaled V chunk's scale 2^e from the fp32 bits of the chunk's abs-max (the device's
 * integer rule; E8M0 byte = e + 127).
 */
size_t qattn_fp8_quant_attention_workspace_bytes(int B, int Hq, int Hkv, int Sq);
int qattn_vblock_exponent(unsigned amax_bits, int out_fmt);
int qattn_fp8_quant_attention_forward(const void* q, const void* k, const void* v, int in_fmt, void* out, void* q8, void* k8,
                                      void* v8, float* scale_q, float* scale_k, float* scale_v, int B, int Hq, int Hkv, int Sq,
                                      int Skv, int D, int fp8_fmt, int scale_mode, int numerics, int is_causal, float sm_scale,
                                      int precision, void* workspace, size_t workspace_bytes, void* stream);

/*
 * The same step with a producer's per-head figures and the optional outputs (head-wise scales only for amax / ssq).
 *   amax_q / amax_k / amax_v  NULL or fp32 [B,Hq] / [B,Hkv] / [B,Hkv]: exact max |x| per head of the 16-bit tensor (what the reference gets
 *       from Inductor fusing `_dynamically_quantize_fp8` into the producer, nn.py:410-418).  A supplied tensor skips the abs-max launch
 *       (amax_v is not needed where V is block-scaled); all supplied: the launch is skipped.  Same bits as the plain call -- under AUTO
 *       when ssq_q / ssq_k come along; with BOTH abs-max and no sums the kernel has no score-spread estimate and wide heads start
 *       one-term (same bound, other bits).  Values enter by magnitude; larger than the true abs-max is safe (coarser scale), smaller clips.
 *   ssq_q / ssq_k  NULL or fp32 [B,Hq] / [B,Hkv] sums of x^2 per head (both or neither); read under AUTO only.
 *   lse / lse_layout  NULL or the per-row log-sum-exp written BY THE SAME LAUNCH as `out` from the kernel's running max and row sum (the
 *       reference: l_vec, tk/attention.py:79,85,333-346,439-452).  Rows of the FP8-MFMA sweep of the D = 128 head-wise kernel carry the
 *       sum of the e4m3-rounded weights the second GEMM consumed -- numerator and denominator see the same weights; `out` does not
 *       change: within 2.5e-2 (natural-log units) of the exact value; every other row within 2e-3 (PATH TABLE column `lse`).
 *   row_path  NULL or one byte per query row [B,Hq,Sq] (QATTN_PATH_*): test / debug output, pre-filled ONE_TERM by a small launch,
 *       overwritten by every other pass for the rows it stores.  No cost when NULL; `out` does not depend on it.
 */
int qattn_fp8_quant_attention_forward_ex(const void* q, const void* k, const void* v, int in_fmt, void* out, void* q8, void* k8,
                                         void* v8, float* scale_q, float* scale_k, float* scale_v, const float* amax_q,
                                         const float* amax_k, const float* amax_v, const float* ssq_q, const float* ssq_k, int B,
                                         int Hq, int Hkv, int Sq, int Skv, int D, int fp8_fmt, int scale_mode, int numerics,
                                         int is_causal, float sm_scale, int precision, float* lse, int lse_layout,
                                         unsigned char* row_path, void* workspace, size_t workspace_bytes, void* stream);

/*
 * PATH TABLE -- which numerics each entry runs.  qattn_describe_path() (host-only) answers from the predicates the dispatch itself uses;
 * tests/test_cpu_boundary.py::test_path_table_matches_dispatch parses these rows and walks them against it.  in_fmt (bf16 / fp16)
 * changes no row; every row accepts e4m3 and e5m2 operands.
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
 * entry: fused = qattn_fp8_quant_attention_forward[_ex]; separate = qattn_fp8_attention_forward / _rowmajor with an fp8 V; separate16 =
 *   the same with v_fmt / pv_fmt 16-bit.  kernel: v2 = hand-scheduled D = 128 (csrc/qattn_attn_v2.hip), v4 = templated
 *   (csrc/qattn_attn_v4.hip), pv16 = csrc/qattn_pv16.h.  q_quant: where Q is quantised (kernel = row by row in the attention prologue,
 *   q8 never written).  v_format: block = one power-of-two (E8M0) scale per 64-key chunk, found inside the quantise pass
 *   (oracle.quantize_v_block); head = one fp32 scale per head.  sweep_p: P of the main sweep (byte = one e4m3 term by the byte
 *   exponential; p16 = 16-bit).  precise: where AUTO's flagged rows / blocks and ACCURATE's blocks go -- v16 = 16-bit P on the caller's
 *   16-bit V (rescued rows with 8 <= R < 24 keep two-term fp8 P on the fp8 V: QATTN_PATH_TWO_TERM); two-term = hi + lo e4m3 P on the fp8
 *   V.  early: query blocks (256 rows) whose first row sees < 1024 keys, every precision mode.  start: what picks a block's STARTING mode
 *   under AUTO (keys = key-count rule + first-chunk forecast; moments = + the pre-pass's per-head sums of squares: a head whose
 *   predicted score variance is >= 1.5 starts on the precise pass).  lse: exact* = asking for the LSE switches the sweep to exact
 *   exponentials (same bound, other output bits); quantised = sums of the e4m3 weights, output bits unchanged.
 * The AUTO bound (2^-6) assumes |v - O| <= ~4.5 for the keys a one-term row keeps (weights < 1/24): it scales with the LARGEST |v|, not
 * V's spread -- V with entries beyond 4-5 standard deviations belongs on QATTN_PRECISION_ACCURATE.
 */
#define QATTN_ENTRY_SEPARATE 0
#define QATTN_ENTRY_SEPARATE_V16 1
#define QATTN_ENTRY_FUSED 2
#define QATTN_KERNEL_V2 0
#define QATTN_KERNEL_V4 1
#define QATTN_KERNEL_PV16 2
#define QATTN_QQUANT_CALLER 0
#define QATTN_QQUANT_PREPASS 1
#define QATTN_QQUANT_KERNEL 2
#define QATTN_VFORMAT_HEAD 0
#define QATTN_VFORMAT_BLOCK 1
#define QATTN_VFORMAT_16BIT 2
#define QATTN_SWEEP_BYTE 0
#define QATTN_SWEEP_EXACT 1
#define QATTN_SWEEP_P16 2
#define QATTN_PRECISE_TWO_TERM 0
#define QATTN_PRECISE_V16 1
#define QATTN_PRECISE_NONE 2
#define QATTN_EARLY_TWO_TERM 0
#define QATTN_EARLY_V16_INLINE 1
#define QATTN_EARLY_V16_LAUNCH 2
#define QATTN_EARLY_NONE 3
#define QATTN_START_KEYS 0
#define QATTN_START_MOMENTS 1
#define QATTN_START_NONE 2
#define QATTN_LSE_SRC_EXACT 0
#define QATTN_LSE_SRC_QUANTISED 1
typedef struct qattn_path_desc {
    int kernel, q_quant, v_format, sweep_p, precise, early, start_mode, lse;
} qattn_path_desc;
/* 0 or a negative code; no device call.  in_fmt: QATTN_FMT_BF16 / _FP16; want_lse: the call passes lse != NULL. */
int qattn_describe_path(int entry, int D, int in_fmt, int scale_mode, int Skv, int want_lse, qattn_path_desc* desc);

/* Measurement aid (bench.py `in_kernel_clock_ghz`): the fused step on an instantiation whose waves bracket their KV sweep with
 * s_memtime / s_memrealtime; `stamps` receives {cycles, 100 MHz ticks} per wave (8 per 256-row block, (b, h, block) order).  Only
 * D = 128, bf16, head-wise, e4m3 (else QATTN_ERR_UNSUPPORTED_FMT, before anything is written).  Product entries execute no stamp. */
size_t qattn_attention_stamp_bytes(int B, int Hq, int Sq);
int qattn_fp8_quant_attention_forward_stamped(const void* q, const void* k, const void* v, int in_fmt, void* out, void* q8, void* k8,
                                              void* v8, float* scale_q, float* scale_k, float* scale_v, int B, int Hq, int Hkv, int Sq,
                                              int Skv, int D, int fp8_fmt, int scale_mode, int numerics, int is_causal, float sm_scale,
                                              int precision, void* workspace, size_t workspace_bytes, void* stamps, size_t stamps_bytes,
                                              void* stream);

/* 16-bit sibling path: the non-fp8 build of the same kernel (tk/attention.py:212,238-240,289-313) behind
 * `quantum_attn::attention_forward` (ops.py:17-45).  q / out row-major bf16 or fp16 (`fmt`); k16 / v16 re-laid by qattn_pack16 into
 * K16FRAG / V16FRAG.  Both GEMMs on v_mfma_f32_32x32x16_{bf16,f16}.  fast_exp = 1: linear-mantissa 2^x (1.8 % rms per weight) for
 * rows that see >= 1024 keys -- only for rows known to be flat.  lse: NULL or dense fp32 [B,Hq,Sq]. */
size_t qattn_16bit_tensor_bytes(int layout, int B, int H, int S, int D);
int qattn_pack16(const void* x_rowmajor, void* x_packed, int B, int H, int S, int D, int out_layout, void* stream);
int qattn_attention_forward_16(const void* q, const void* k16, const void* v16, void* out, float* lse, int B, int Hq,
                               int Hkv, int Sq, int Skv, int D, int fmt, int is_causal, float sm_scale, int fast_exp,
                               void* stream);

/* Measurement aids for bench.py (not part of the drop-in surface).  qattn_profile_attention(1): every following attention launch of
 * the calling thread is bracketed by two HIP events on its own stream; qattn_last_attention_ms() returns the time between them
 * (synchronises on the second), negative when off.  qattn_mfma_probe: a bare v_mfma_f32_32x32x64_f8f6f4 loop (operands in registers,
 * two waves per SIMD, one workgroup per CU) on the fp8 bytes in the first 64 KiB of `scratch`; the rest of `scratch` receives
 * {cycles, 100 MHz ticks} per wave; *flops_per_launch = iters x 4 x waves x 2 x 32 x 32 x 64. */
void qattn_profile_attention(int enable);
float qattn_last_attention_ms(void);
size_t qattn_mfma_probe_bytes(void);
int qattn_mfma_probe(void* scratch, size_t scratch_bytes, int iters, double* flops_per_launch, int* waves, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QATTN_H_ */
