// Probe 2: how much VALU work hides beside v_mfma_f32_32x32x64_f8f6f4 with 1 and 2 waves per SIMD.
// Each wave runs ITER x { 1 MFMA ; K independent VALU ops of one kind }.  Whole-kernel time -> cycles per MFMA slot.
// Test infrastructure only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// KIND: 0 fma, 1 exp2, 2 cvt_pk_fp8, 3 mixed softmax-like (per 4 elements: 4 fma, 4 exp, 2 max3-ish, 4 add, 2 cvt)
template <int KIND, int K, bool MFMA>
__global__ __launch_bounds__(512) void k_mix(float* out, int iters, float seed, unsigned long long* cyc) {
  v8i a, b;
  for (int i = 0; i < 8; i++) { a[i] = 0x38383838 + (threadIdx.x * (i + 1)) % 5; b[i] = 0x30303030 + i; }
  v16f acc0 = {0}, acc1 = {0};
  float x[16];
  for (int i = 0; i < 16; i++) x[i] = seed + threadIdx.x * 1e-3f + i * 0.01f;
  int r[4] = {0, 0, 0, 0};
  float mx = 0.f, sm = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (MFMA) {
        if (u & 1) acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc1, 0, 0, 0, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc0, 0, 0, 0, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < K; j++) {
        const int i = (u * K + j) & 15;
        if (KIND == 0) x[i] = __builtin_fmaf(x[i], seed, 0.25f);
        if (KIND == 1) x[i] = __builtin_amdgcn_exp2f(x[i]);
        if (KIND == 2) { r[j & 3] = __builtin_amdgcn_cvt_pk_fp8_f32(x[i], x[(i + 1) & 15], r[j & 3], false); asm volatile("" : "+v"(x[i])); }
        if (KIND == 3) {  // one "element group": fma, exp, max, add  (+ a cvt every other)
          float t = __builtin_amdgcn_exp2f(__builtin_fmaf(x[i], seed, 0.25f));
          mx = __builtin_fmaxf(mx, x[i]);
          sm += t;
          if (j & 1) r[j & 3] = __builtin_amdgcn_cvt_pk_fp8_f32(t, sm, r[j & 3], false);
          x[i] = t;
        }
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) atomicAdd(cyc, t1 - t0);
  float s = mx + sm;
  for (int i = 0; i < 16; i++) s += x[i] + acc0[i] + acc1[i];
  for (int i = 0; i < 4; i++) s += (float)r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int K, bool MFMA>
static double run(int threads, double* cycles_per_slot) {
  int blocks = 256; float* out; CK(hipMalloc(&out, blocks * threads * 4));
  unsigned long long* cyc; CK(hipMalloc(&cyc, 8));
  int iters = 4000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_mix<KIND, K, MFMA>), blocks, threads, 0, 0, out, 200, 0.999f, cyc);
  CK(hipMemset(cyc, 0, 8));
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_mix<KIND, K, MFMA>), blocks, threads, 0, 0, out, iters, 0.999f, cyc);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  *cycles_per_slot = (double)c / ((double)blocks * (threads / 64)) / ((double)iters * 4);
  CK(hipFree(out)); CK(hipFree(cyc));
  return (double)ms * 1e-3 / ((double)iters * 4);   // seconds per (MFMA + K ops) slot per wave
}

template <int KIND, int K>
static void row(const char* name) {
  double c1, c2, d1, d2;
  double m1 = run<KIND, K, true>(256, &c1), m2 = run<KIND, K, true>(512, &c2), v1 = run<KIND, K, false>(256, &d1), v2 = run<KIND, K, false>(512, &d2);
  // cX/dX = mean per-wave s_memtime cycles per slot; clock = cycles / wall
  printf("%-8s K=%2d | MFMA+VALU 1w: %.1f cyc/slot @%.2fGHz | 2w: %.1f cyc/slot/wave @%.2fGHz (SIMD: %.1f per 2 slots) | VALU-only 1w: %.2f cyc/op  2w: %.2f cyc/op/wave\n",
         name, K, c1, c1 / m1 / 1e9, c2, c2 / m2 / 1e9, c2, K ? d1 / K : 0.0, K ? d2 / K : 0.0);
}

int main() {
  printf("slot = 1 MFMA(32x32x64 fp8, 64 cyc) + K VALU ops; mean per-wave s_memtime cycles\n");
  row<0, 0>("none");
  row<0, 4>("fma"); row<0, 8>("fma"); row<0, 12>("fma"); row<0, 16>("fma"); row<0, 24>("fma"); row<0, 32>("fma");
  row<1, 2>("exp2"); row<1, 4>("exp2"); row<1, 6>("exp2"); row<1, 8>("exp2"); row<1, 12>("exp2");
  row<2, 4>("cvtfp8"); row<2, 8>("cvtfp8");
  row<3, 2>("softmx"); row<3, 4>("softmx"); row<3, 6>("softmx"); row<3, 8>("softmx");
  return 0;
}
