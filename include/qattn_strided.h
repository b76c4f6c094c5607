/*
 * qattn_strided.h -- the fused entry of include/qattn.h on STRIDED VIEWS of the 16-bit inputs (ABI 8).
 *
 * Attention inputs usually reach the reference as views: q = x.view(B, S, H, D).transpose(1, 2), or slices of a packed [B,S,3,H,D] QKV
 * projection.  The reference reads such q / k in its Inductor-generated quantiser (nn.py:14-19 under torch.compile, nn.py:484-488) and COPIES
 * such a v in its launcher (`.contiguous()`, tk/attention.py:419-421, 455-456).  Here every kernel that touches the 16-bit tensors -- the
 * abs-max pass, the quantise pass, the attention kernels' in-kernel Q quantisation and their 16-bit-V passes -- takes the strides: nothing is
 * copied, and out / lse / row_path / scale_* are, bit for bit, those of qattn_fp8_quant_attention_forward_ex on dense copies of the views.
 *
 *   strides   12 element strides: {batch, head, row} of q, then of k, of v and of out, for tensors indexed [b][h][s][d].  D is innermost and
 *             dense (stride 1).  Every stride a non-negative multiple of 8 (rows 16-byte aligned), row stride in [D, 2^23]; q, k, v, out 16-byte
 *             aligned; else QATTN_ERR_INVALID_ARG.  NULL = dense [B,H,S,D] (= ..._forward_ex).  A stride of 0 broadcasts an INPUT (e.g. one
 *             K / V for every batch).  `out` as a view: e.g. the transpose of a dense [B,Sq,Hq,D] buffer, which the caller reshapes to
 *             [B,Sq,Hq D] for its output projection without a copy (the reference allocates a dense [B,Hq,Sq,D]: tk/attention.py:434-437).
 *   everything else: as qattn_fp8_quant_attention_forward_ex.
 */
#ifndef QATTN_STRIDED_H_
#define QATTN_STRIDED_H_

#include "qattn.h"

#ifdef __cplusplus
extern "C" {
#endif

int qattn_fp8_quant_attention_forward_strided(const void* q, const void* k, const void* v, const long long* strides, int in_fmt, void* out,
                                              void* q8, void* k8, void* v8, float* scale_q, float* scale_k, float* scale_v,
                                              const float* amax_q, const float* amax_k, const float* amax_v, const float* ssq_q,
                                              const float* ssq_k, int B, int Hq, int Hkv, int Sq, int Skv, int D, int fp8_fmt, int scale_mode,
                                              int numerics, int is_causal, float sm_scale, int precision, float* lse, int lse_layout,
                                              unsigned char* row_path, void* workspace, size_t workspace_bytes, void* stream);

/* The 16-bit sibling path (qattn_pack16 / qattn_attention_forward_16 of qattn.h) on views: qattn_pack16_strided: `strides` = the three
 * element strides {batch, head, row} of x; qattn_attention_forward_16_strided: six, of q then of out; same rules, NULL = dense.
 * k16 / v16 are the K16FRAG / V16FRAG images qattn_pack16[_strided] produced. */
int qattn_pack16_strided(const void* x, const long long* strides, void* x_packed, int B, int H, int S, int D, int out_layout, void* stream);
int qattn_attention_forward_16_strided(const void* q, const long long* strides, const void* k16, const void* v16, void* out, float* lse,
                                       int B, int Hq, int Hkv, int Sq, int Skv, int D, int fmt, int is_causal, float sm_scale,
                                       int fast_exp, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QATTN_STRIDED_H_ */
