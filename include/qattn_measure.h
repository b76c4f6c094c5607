/*
 * qattn_measure.h -- MEASUREMENT entries of libqattn_hip.so (bench.py; no reference counterpart, not part of the drop-in surface of
 * include/qattn.h).  The attention path never calls them; no environment variable changes results.
 */
#ifndef QATTN_MEASURE_H_
#define QATTN_MEASURE_H_

#include "qattn.h"

#ifdef __cplusplus
extern "C" {
#endif

/* bench.py `in_kernel_clock_ghz`: the same step as qattn_fp8_quant_attention_forward on an instantiation of the attention kernel in which
 * every wave brackets its KV sweep with the shader-cycle counter (s_memtime) and the 100 MHz real-time counter (s_memrealtime).  `stamps`
 * receives {cycles, ticks} per wave, 8 waves per 256-row query block, blocks in (b, h, block) order (qattn_attention_stamp_bytes() bytes):
 * cycles / ticks x 0.1 = the clock in GHz the chip held inside the kernel.  Only D = 128, bf16, head-wise, e4m3 -- else
 * QATTN_ERR_UNSUPPORTED_FMT before anything is written; outputs are those of the unstamped step.  Product entries execute no stamp. */
size_t qattn_attention_stamp_bytes(int B, int Hq, int Sq);
int qattn_fp8_quant_attention_forward_stamped(const void* q, const void* k, const void* v, int in_fmt, void* out, void* q8, void* k8,
                                              void* v8, float* scale_q, float* scale_k, float* scale_v, int B, int Hq, int Hkv, int Sq,
                                              int Skv, int D, int fp8_fmt, int scale_mode, int numerics, int is_causal, float sm_scale,
                                              int precision, void* workspace, size_t workspace_bytes, void* stamps, size_t stamps_bytes,
                                              void* stream);

/* bench.py `roofline.achieved`: qattn_profile_attention(1) makes every following attention launch of the calling thread be bracketed by two
 * HIP events on its own stream; qattn_last_attention_ms() returns the milliseconds between them for the most recent launch (it
 * synchronises on the second), negative when profiling is off.
 * bench.py `roofline.practical_peak`: qattn_mfma_probe runs a bare v_mfma_f32_32x32x64_f8f6f4 loop -- operands in registers, four
 * independent accumulators per wave, two waves per SIMD, one 512-thread workgroup per CU -- on the fp8 bytes in the first 64 KiB of
 * `scratch` (random e4m3 bytes for a figure comparable with the attention kernel's; constant bytes read 30-40 % higher: the chip holds a
 * higher clock on them).  The rest of `scratch` (qattn_mfma_probe_bytes() in all) receives {cycles, 100 MHz ticks} of every wave's loop;
 * *flops_per_launch = iters x 4 x waves x 2 x 32 x 32 x 64; the caller times the launches. */
void qattn_profile_attention(int enable);
float qattn_last_attention_ms(void);
size_t qattn_mfma_probe_bytes(void);
int qattn_mfma_probe(void* scratch, size_t scratch_bytes, int iters, double* flops_per_launch, int* waves, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QATTN_MEASURE_H_ */
