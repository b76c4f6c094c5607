/*
 * qattn.h -- C ABI of libqattn_hip.so: MI355X (gfx950) FP8 fused attention forward + bf16/fp16->fp8 quant pre-pass.
 *
 * This is the drop-in boundary for the ONE hot path of WaveSpeedAI/QuantumAttention (reference @ 2025-02-22):
 *
 *   qattn_fp8_attention_forward  replaces  the pybind entry `attention_forward(q, k, v, scale_q, scale_k, causal)`
 *                                          (src/quantum_attn/tk/attention.py:355-360, 688-702) behind the custom op
 *                                          `quantum_attn::fp8_attention_forward` (src/quantum_attn/ops.py:98-121),
 *                                          i.e. kernel `fwd_attend_ker<D,causal,..>` (tk/attention.py:97-349).
 *   qattn_quant_fp8              replaces  `_dynamically_quantize_fp8` (src/quantum_attn/nn.py:14-19) as invoked by
 *                                          `_fp8_attention_wrapper` (nn.py:410-418) / `dynamically_quantize_fp8`
 *                                          (nn.py:22-42), which the reference leaves to Inductor-generated Triton.
 *   qattn_pack_fp8               (no reference counterpart) re-lays a row-major fp8 K or V tensor into the MFMA
 *                                          fragment layouts below; used when the caller hands pre-quantised q/k to the
 *                                          op (ops.py:98-110 accepts float8 query/key with scale_q/scale_k).
 *
 * Conventions (same as the reference launcher, tk/attention.py:362-465, unless noted):
 *   - plain C, no torch types; every pointer is a DEVICE pointer on the current HIP device;
 *   - tensors are dense [B, H, S, D] ("row-major") unless a fragment layout is named;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls enqueue work and return
 *     immediately -- no host synchronisation, no allocation, graph-capture safe.  In stream order a call begins after everything
 *     queued on `stream` before it and ends before everything queued after it; inside, a causal attention call on the templated
 *     kernel (D = 64 / 256, token-wise scales, fp16 inputs) runs the launch of its early rows on ONE internal non-blocking stream per
 *     host thread and device, forked from and joined to `stream` with events (two parallel branches under capture).  That stream
 *     and its two events are the one exception to "no allocation": they are created on the thread's first such call outside a
 *     capture (a capturing call before that runs its launches one after the other; if the creation fails, every later call of the
 *     thread stays on `stream`) and are destroyed when the host thread exits;
 *   - return 0 on success or a negative QATTN_ERR_* code; nothing is thrown; qattn_strerror() names the code.
 *
 * Fragment layouts (private to this library; produced by qattn_quant_fp8 / qattn_pack_fp8, consumed by the
 * attention kernel; sequence length padded with zero bytes to a multiple of 64 keys, Sp = 64*ceil(S/64)):
 *   QATTN_LAYOUT_KFRAG  per (b,h): chunks of 64 keys; chunk = [t:2][s:D/64][hh:2][half:2][key:32][16 bytes],
 *                       byte j of a 16-byte piece = K[64*chunk + 32*t + key][64*s + 32*hh + 16*half + j].
 *   QATTN_LAYOUT_VFRAG  per (b,h): chunks of 64 keys; chunk = [m:D/32][hh:2][half:2][d:32][16 bytes],
 *                       byte 4*w+i of a piece = V[64*chunk + 32*half + 8*w + 4*hh + i][32*m + d]   (w,i in 0..3).
 *   Both are D*Sp bytes per (b,h): the A operands of v_mfma_f32_32x32x64_f8f6f4 for S^T = K.Q^T and O^T = V^T.P^T,
 *   conflict-free for ds_read_b128 and linear for LDS-DMA.
 */
#ifndef QATTN_H_
#define QATTN_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QATTN_ABI_VERSION 6   /* 6 (round 4): v_fmt = QATTN_FMT_BF16 / _FP16 in qattn_fp8_attention_forward, qattn_mfma_probe */

/* element formats */
#define QATTN_FMT_E4M3 0 /* OCP float8_e4m3fn  (torch.float8_e4m3fn) */
#define QATTN_FMT_E5M2 1 /* OCP float8_e5m2    (torch.float8_e5m2)   */
#define QATTN_FMT_BF16 2
#define QATTN_FMT_FP16 3

/* scale granularity: head-wise = one fp32 per (b,h) [B,H]; token-wise = one per row [B,H,S]  (nn.py:410-414) */
#define QATTN_SCALE_HEAD 0
#define QATTN_SCALE_TOKEN 1

/* memory layouts of an fp8 tensor */
#define QATTN_LAYOUT_ROWMAJOR 0
#define QATTN_LAYOUT_KFRAG 1
#define QATTN_LAYOUT_VFRAG 2
/* fragment layouts of a 16-bit (bf16/fp16) K or V tensor for the 16-bit sibling path (16-byte pieces of 8 elements):
 *   QATTN_LAYOUT_K16FRAG chunk = [t:2][s:D/16][hh:2][key:32][8 elts], piece = K[64c + 32t + key][16s + 8hh + (0..7)]
 *   QATTN_LAYOUT_V16FRAG chunk = [t:2][m:D/32][s:2][hh:2][d:32][8 elts],
 *                        element j of a piece = V[64c + 32t + 16s + 8(j>>2) + 4hh + (j&3)][32m + d]               */
#define QATTN_LAYOUT_K16FRAG 3
#define QATTN_LAYOUT_V16FRAG 4

/* quantiser numerics: 0 = what the reference's compiled GPU path computes (fp32 scale,
 * quotient rounded to the input dtype), 1 = the reference's eager arithmetic (everything in the input dtype). */
#define QATTN_NUMERICS_COMPILED 0
#define QATTN_NUMERICS_EAGER 1

/* how the softmax probabilities P enter the second fp8 GEMM.  The reference keeps P (and V) in 16 bit
 * (src/quantum_attn/tk/attention.py:72,286,318); here P is e4m3, and ONE e4m3 term (3 mantissa bits) is accurate enough
 * only for rows whose weight is spread over many keys:
 *   AUTO     (default) one-term P, checked per row: a 256-row query block in which some row's largest softmax weight
 *            exceeds 1/24 (R = l / p_max < 24; or whose effective key count is below 192) has those rows (or, if many, the
 *            whole block) recomputed with more precision.
 *   FAST     one-term P wherever a row sees >= 1024 keys (what other fp8 attention kernels do); no check.
 *   ACCURATE the precise pass everywhere (~bf16-P accuracy, 1.5x the matrix work).
 * "More precision" is two-term (hi + lo) e4m3 P on the fp8 V -- or, in the fused entry (qattn_fp8_quant_attention_forward, which
 * has the caller's 16-bit V at hand) at D = 128 with head-wise scales, the reference kernel's own numerics: 16-bit P on the original
 * 16-bit V (whole blocks; rescued rows with a weight above 1/8).  In every mode query blocks that see fewer than 1024 keys
 * (short sequences, early causal rows) take the precise pass. */
#define QATTN_PRECISION_AUTO 0
#define QATTN_PRECISION_FAST 1
#define QATTN_PRECISION_ACCURATE 2

/* layout / convention of the optional log-sum-exp output:
 *   NATURAL   dense fp32 [B,Hq,Sq], ln sum_j exp(score_j) of the scaled scores
 *   REFERENCE the vector the reference defines in its (disabled) epilogue, tk/attention.py:333-346,439-446:
 *             L = -(ln l + m ln2) * sqrt(D) = -sqrt(D) * NATURAL, rows of consecutive (b,h) spaced
 *             qattn_lse_row_stride(Sq, REFERENCE) = ceil(Sq*4/16)*16/4 floats apart (row padded to 16 bytes).
 *             (The reference's constants are -8 for D = 64 and -11.3137 = -sqrt(128) for every other D; here -sqrt(D).) */
#define QATTN_LSE_NATURAL 0
#define QATTN_LSE_REFERENCE 1

/* error codes */
#define QATTN_OK 0
#define QATTN_ERR_INVALID_ARG (-1)     /* NULL pointer, non-positive dimension, unknown enum */
#define QATTN_ERR_UNSUPPORTED_DIM (-2) /* head_dim not in {64,128,256} (nn.py:45-49), or Hq % Hkv != 0 */
#define QATTN_ERR_UNSUPPORTED_FMT (-3) /* format / layout combination not implemented */
#define QATTN_ERR_WORKSPACE (-4)       /* workspace too small */
#define QATTN_ERR_LAUNCH (-5)          /* HIP launch failed (hipGetLastError != hipSuccess) */
#define QATTN_ERR_DEVICE (-6)          /* current device is not gfx950 */

int qattn_abi_version(void);
const char* qattn_strerror(int code);

/* 0 if the current HIP device is gfx950 (MI355X), else QATTN_ERR_DEVICE.  Replaces the reference's
 * `cuda_capability_compare("ge", 9, 0)` gate (src/quantum_attn/utils/checks.py:57-64, nn.py:214). */
int qattn_check_device(void);

/* bytes of an fp8 tensor [B,H,S,D] stored in `layout` (fragment layouts pad S to a multiple of 64). */
size_t qattn_fp8_tensor_bytes(int layout, int B, int H, int S, int D);

/* bytes of scratch qattn_quant_fp8 needs for these arguments (0 for token-wise). */
size_t qattn_quant_workspace_bytes(int B, int H, int S, int D, int scale_mode);

/*
 * Quant pre-pass (nn.py:14-19): scale = clamp_min(amax|x| * (1/fmax), eps_f32); x8 = fp8(clamp(x/scale, +-fmax)).
 *   x         [B,H,S,D] bf16 or fp16 (in_fmt), dense
 *   x8        fp8 payload in `out_layout` (qattn_fp8_tensor_bytes bytes), out_fmt = E4M3 (reference) or E5M2
 *   scale     fp32 [B,H] (QATTN_SCALE_HEAD, amax over S and D) or [B,H,S] (QATTN_SCALE_TOKEN, amax over D)
 *   workspace device scratch of qattn_quant_workspace_bytes() bytes (may be NULL when that is 0)
 * Bit-exact to the reference for numerics = QATTN_NUMERICS_COMPILED / _EAGER respectively.
 */
int qattn_quant_fp8(const void* x, int in_fmt, void* x8, float* scale, int B, int H, int S, int D, int out_fmt,
                    int scale_mode, int numerics, int out_layout, void* workspace, size_t workspace_bytes,
                    void* stream);

/*
 * Fused pre-pass of one attention call: quantises q, k and v in ONE amax launch + ONE quantise launch
 * (q8 row-major, k8 QATTN_LAYOUT_KFRAG, v8 QATTN_LAYOUT_VFRAG; q and k scaled per `scale_mode`, v always head-wise).
 * Same numerics as three qattn_quant_fp8 calls.  `workspace` needs qattn_quant_qkv_workspace_bytes() bytes (256 per-block
 * abs-max words per head of q, k, v + 256 per-block sums of squares per head of q and k, which the fused entry below fills
 * for its attention kernel: 1 KiB per head each; nothing is zeroed, the consumers reduce the blocks' entries themselves).
 * This is what `_fp8_attention_wrapper` does for its two tensors at nn.py:410-418, plus the build's quantised V.
 */
size_t qattn_quant_qkv_workspace_bytes(int B, int Hq, int Hkv);
int qattn_quant_qkv_fp8(const void* q, const void* k, const void* v, int in_fmt, void* q8, void* k8, void* v8,
                        float* scale_q, float* scale_k, float* scale_v, int B, int Hq, int Hkv, int Sq, int Skv, int D,
                        int out_fmt, int scale_mode, int numerics, void* workspace, size_t workspace_bytes, void* stream);

/* Re-lay a dense row-major fp8 tensor [B,H,S,D] into QATTN_LAYOUT_KFRAG or QATTN_LAYOUT_VFRAG (byte permutation). */
int qattn_pack_fp8(const void* x8_rowmajor, void* x8_packed, int B, int H, int S, int D, int out_layout, void* stream);

/*
 * FP8 fused attention forward:  O = softmax(sm_scale * (sq*Q8)(sk*K8)^T [+causal mask]) (sv*V8),  flash-style.
 *   q8        [B,Hq,Sq,D]   fp8 (qk_fmt), row-major
 *   k8        [B,Hkv,Skv,D] fp8 (qk_fmt), QATTN_LAYOUT_KFRAG
 *   v8        [B,Hkv,Skv,D] fp8 (v_fmt = qk_fmt), QATTN_LAYOUT_VFRAG -- both GEMMs on FP8 MFMA;
 *             or (v_fmt = QATTN_FMT_BF16 / QATTN_FMT_FP16 = out_fmt) the ORIGINAL 16-bit value tensor, dense ROW-MAJOR, no scale_v:
 *             every row then runs the reference kernel's own P.V numerics -- FP8 QK^T, 16-bit P, 16-bit V (tk/attention.py:72,286,318) --
 *             on v_mfma_f32_32x32x16_{bf16,f16} (csrc/qattn_pv16.h; about 1.5x the time; `precision` plays no part: P carries 8 / 11
 *             mantissa bits).  Every supported head dim (64 / 128 / 256), head- and token-wise scales, e4m3 / e5m2 q and k.
 *   out       [B,Hq,Sq,D]   bf16 or fp16 (out_fmt), row-major, written in full
 *   lse       NULL, or fp32 log-sum-exp of the scaled scores per query row in `lse_layout` (B*Hq rows of
 *             qattn_lse_row_stride(Sq, lse_layout) floats) -- the per-row vector the reference defines but disables
 *   scale_q   fp32 [B,Hq] (head-wise) or [B,Hq,Sq] (token-wise);  scale_k likewise with Hkv,Skv
 *   scale_v   fp32 [B,Hkv] or NULL (= 1.0)
 *   sm_scale  softmax scale; <= 0 selects 1/sqrt(D) (the reference hard-wires it, tk/attention.py:208-210)
 *   is_causal keep key j <= query i (aten top-left alignment; the reference requires Sq == Skv, tests/test_interface.py:32)
 *   precision QATTN_PRECISION_*
 *   workspace device scratch of qattn_attention_workspace_bytes(B, Hq, Sq) bytes: the block hand-out counters of a causal
 *             launch, or of a non-causal one with many query blocks per CU (zeroed by the call itself with a small kernel --
 *             graph-capture safe) and one word per 32-row query group.  Needed for QATTN_PRECISION_AUTO; may be NULL otherwise,
 *             a causal launch then uses one workgroup per query block and a large non-causal one equal static shares (a few
 *             per cent slower on long sequences; the same results bit for bit).  Nothing in it outlives the call.
 * Both GEMMs run on v_mfma_f32_32x32x64_f8f6f4 (fp8 V); accumulation, running max/sum and the softmax are fp32.
 */
size_t qattn_attention_workspace_bytes(int B, int Hq, int Sq);
size_t qattn_lse_row_stride(int Sq, int lse_layout);
int qattn_fp8_attention_forward(const void* q8, const void* k8, const void* v8, void* out, float* lse,
                                const float* scale_q, const float* scale_k, const float* scale_v, int B, int Hq,
                                int Hkv, int Sq, int Skv, int D, int qk_fmt, int v_fmt, int out_fmt, int scale_mode,
                                int is_causal, float sm_scale, int precision, int lse_layout, void* workspace,
                                size_t workspace_bytes, void* stream);

/*
 * The whole step of `_fp8_attention_wrapper` for 16-bit inputs (nn.py:394-430: quantise q and k, then the fp8 op) in one
 * call: pre-pass (qattn_quant_qkv_fp8 semantics) + attention (qattn_fp8_attention_forward semantics, no LSE) on `stream`.
 * q8 / k8 / v8 / scale_* are caller-provided outputs+scratch with the sizes qattn_quant_qkv_fp8 documents; `workspace` needs
 * qattn_fp8_quant_attention_workspace_bytes().  Where the attention kernel can quantise its own Q rows (D = 128, bf16 or fp16,
 * head-wise) the pre-pass skips Q's payload -- q8 is then left untouched, scale_q is still written -- which saves one read
 * and one write of Q.  For every head dim, 16-bit input format and scale mode the query blocks (256 rows) whose first row sees fewer
 * than 1024 keys -- early causal rows, every row of a short sequence -- attend the ORIGINAL 16-bit V with 16-bit P (the reference's
 * numerics, as v_fmt = 16-bit above) instead of the quantised V: inside the fused kernel for head-wise D = 128 inputs (bf16 and fp16), by a launch of
 * their own otherwise; on those rows the step's results are NOT those of the separate calls (which only have the fp8 V).  There, and with head-wise scales at D = 64 / 256 (both for Skv <= 16384), V is also quantised
 * differently from qattn_quant_qkv_fp8: one power-of-two scale per 64-key chunk, found inside the quantise pass (no abs-max pass over V) and applied by the kernel's PV products as
 * the MFMA's E8M0 block scale; v8 then holds those payloads, scale_v is written as 1.0 and the chunk scales live in the
 * workspace (oracle restatement: oracle.quantize_v_block).  Everywhere else results are bit-identical to the separate
 * calls, with one more documented exception: under
 * QATTN_PRECISION_AUTO (head-wise, D = 128) the pre-pass also hands the attention kernel every head's sum of squares, and a
 * head whose predicted score variance is >= 1.5 starts on the precise pass (what QATTN_PRECISION_ACCURATE computes for it) instead
 * of being swept once with one-term P first; heads below that -- N(0,1)-like data -- take the same decisions as the separate calls.
 */
size_t qattn_fp8_quant_attention_workspace_bytes(int B, int Hq, int Hkv, int Sq);
/* The exponent e of the block-scaled V's chunk scale 2^e, from the fp32 bits of the chunk's abs-max (host function, the same
 * integer rule the quantise pass applies on the device; out_fmt: QATTN_FMT_E4M3 / QATTN_FMT_E5M2).  The E8M0 byte is e + 127. */
int qattn_vblock_exponent(unsigned amax_bits, int out_fmt);
int qattn_fp8_quant_attention_forward(const void* q, const void* k, const void* v, int in_fmt, void* out, void* q8, void* k8,
                                      void* v8, float* scale_q, float* scale_k, float* scale_v, int B, int Hq, int Hkv, int Sq,
                                      int Skv, int D, int fp8_fmt, int scale_mode, int numerics, int is_causal, float sm_scale,
                                      int precision, void* workspace, size_t workspace_bytes, void* stream);

/*
 * The same step for a caller that PRODUCES q / k / v itself and already knows per-head figures of them (the epilogue of its
 * projection or RoPE kernel): the counterpart of what the reference gets from Inductor, which traces `_dynamically_quantize_fp8`
 * into the caller's graph and fuses the abs-max reduction with whatever wrote q and k (nn.py:410-418).  Head-wise scales only.
 *   amax_q / amax_k / amax_v  NULL, or fp32 [B,Hq] / [B,Hkv] / [B,Hkv]: max |x| over each head of the 16-bit tensor, exactly (the
 *                             fp32 value of the largest 16-bit magnitude).  A tensor with a supplied abs-max takes no part in the
 *                             abs-max launch; with all of them supplied (amax_v is not needed where V is block-scaled: head-wise
 *                             scales, Skv <= 16384, every head dim, bf16 and fp16 inputs) the launch is skipped -- at B4 H32 S4096 D128 that is 0.05 of 0.64 ms.  The
 *                             results are bit-identical to qattn_fp8_quant_attention_forward's -- under QATTN_PRECISION_AUTO when
 *                             ssq_q / ssq_k come along (below), except for a head whose estimated score variance sits exactly on
 *                             the dead-band edge (1.5): the caller's fp32 sums differ from the pass's partial sums in the last
 *                             bits and may start such a head in the other mode (same bound).  Without the sums: with only ONE of
 *                             amax_q / amax_k supplied both tensors still go through the abs-max pass for their sums of squares
 *                             (nothing is saved, the plain call's bits); with BOTH supplied the pass is skipped, the kernel has no
 *                             score-spread estimate and heads with a wide spread start one-term as in the separate calls: the same
 *                             bound, other bits.  Preconditions: every value finite; it enters by magnitude (the sign bit is
 *                             dropped).  A value LARGER than the true abs-max is safe (a coarser scale, no clipping) but no longer
 *                             the reference's scale; a smaller one clips, a NaN makes that head's scale NaN.
 *   ssq_q / ssq_k             NULL, or fp32 [B,Hq] / [B,Hkv]: sum of x^2 over each head (both or neither).  Only read under
 *                             QATTN_PRECISION_AUTO, where the pre-pass otherwise accumulates them for the score-spread estimate
 *                             that picks a head's starting precision; without them (and without the abs-max pass over q and k)
 *                             every head starts in one-term mode, as through the separate calls.
 */
int qattn_fp8_quant_attention_forward_ex(const void* q, const void* k, const void* v, int in_fmt, void* out, void* q8, void* k8,
                                         void* v8, float* scale_q, float* scale_k, float* scale_v, const float* amax_q,
                                         const float* amax_k, const float* amax_v, const float* ssq_q, const float* ssq_k, int B,
                                         int Hq, int Hkv, int Sq, int Skv, int D, int fp8_fmt, int scale_mode, int numerics,
                                         int is_causal, float sm_scale, int precision, void* workspace, size_t workspace_bytes,
                                         void* stream);

/*
 * Measurement aid (bench.py `in_kernel_clock_ghz`): the same step as qattn_fp8_quant_attention_forward, run on an instantiation of the
 * attention kernel in which every wave brackets its KV sweep with the shader-cycle counter (s_memtime) and the 100 MHz real-time
 * counter (s_memrealtime).  `stamps` receives {cycles, ticks} per wave, 8 waves per 256-row query block, blocks in (b, h, block)
 * order: qattn_attention_stamp_bytes() bytes.  cycles / ticks x 0.1 is the clock in GHz the chip held inside the kernel.  Only where
 * the hand-scheduled kernel runs the fused step (D = 128, bf16, head-wise, e4m3), else QATTN_ERR_UNSUPPORTED_FMT; outputs are those
 * of the unstamped step.  The product entries execute no stamp.
 */
size_t qattn_attention_stamp_bytes(int B, int Hq, int Sq);
int qattn_fp8_quant_attention_forward_stamped(const void* q, const void* k, const void* v, int in_fmt, void* out, void* q8, void* k8,
                                              void* v8, float* scale_q, float* scale_k, float* scale_v, int B, int Hq, int Hkv, int Sq,
                                              int Skv, int D, int fp8_fmt, int scale_mode, int numerics, int is_causal, float sm_scale,
                                              int precision, void* workspace, size_t workspace_bytes, void* stamps, size_t stamps_bytes,
                                              void* stream);

/*
 * 16-bit sibling path: the non-fp8 build of the same kernel (TK_ATTN_IS_FP8 undefined, tk/attention.py:212,238-240,
 * 289-313) behind `quantum_attn::attention_forward(query, key, value, scale=None, is_causal=False)`
 * (src/quantum_attn/ops.py:17-45).  q/out row-major [B,Hq,Sq,D] bf16 or fp16 (`fmt`); k16/v16 are the key/value
 * tensors re-laid by qattn_pack16 into QATTN_LAYOUT_K16FRAG / QATTN_LAYOUT_V16FRAG (qattn_16bit_tensor_bytes bytes).
 * D in {64,128,256}.  Both GEMMs run on v_mfma_f32_32x32x16_{bf16,f16}; softmax, running max/sum and accumulation fp32.
 * fast_exp = 0 (default): exact exp2 as the reference; 1: linear-mantissa 2^x (1.8 % rms error per weight) for rows that
 * see >= 1024 keys -- only for rows known to be flat.  lse: NULL or dense fp32 [B,Hq,Sq] (QATTN_LSE_NATURAL).
 */
size_t qattn_16bit_tensor_bytes(int layout, int B, int H, int S, int D);
int qattn_pack16(const void* x_rowmajor, void* x_packed, int B, int H, int S, int D, int out_layout, void* stream);
int qattn_attention_forward_16(const void* q, const void* k16, const void* v16, void* out, float* lse, int B, int Hq,
                               int Hkv, int Sq, int Skv, int D, int fmt, int is_causal, float sm_scale, int fast_exp,
                               void* stream);

/* Measurement aid for bench.py (not part of the drop-in surface).  qattn_profile_attention(1) makes every following
 * attention launch on the calling thread's current device be bracketed by two HIP events on its own stream;
 * qattn_last_attention_ms() returns the milliseconds between them for the most recent launch (it synchronises on the
 * second event), or a negative value when profiling is off.  Off by default; no environment variable changes results. */
void qattn_profile_attention(int enable);
float qattn_last_attention_ms(void);

/*
 * Measurement aid (bench.py `roofline.practical_peak`): a bare v_mfma_f32_32x32x64_f8f6f4 loop -- operands in registers, four
 * independent accumulators per wave, two waves per SIMD, one 512-thread workgroup per CU -- on the fp8 bytes the caller put into
 * the first 64 KiB of `scratch` (random e4m3 bytes for a figure comparable with the attention kernel's; constant bytes read
 * 30-40 % higher because the chip holds a higher clock on them).  The rest of `scratch` (qattn_mfma_probe_bytes() in all) receives
 * {shader cycles, 100 MHz ticks} of every wave's loop: cycles / ticks x 0.1 = the clock in GHz inside the loop.  *flops_per_launch
 * = iters x 4 x waves x 2 x 32 x 32 x 64; the caller times the launches (HIP events on `stream`).  No reference counterpart; the
 * attention path never calls it.
 */
size_t qattn_mfma_probe_bytes(void);
int qattn_mfma_probe(void* scratch, size_t scratch_bytes, int iters, double* flops_per_launch, int* waves, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QATTN_H_ */
