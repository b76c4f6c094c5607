// Probe 4: attainable v_mfma_f32_32x32x64_f8f6f4 rate on RANDOM operands (DVFS: the clock the chip holds depends on data).
// Each wave loops over MFMAs with per-lane random fp8 A/B operands held in registers (rotated every iteration so the
// compiler cannot hoist), 4 independent accumulators.  Reports TFLOP/s and the in-kernel clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ __launch_bounds__(512) void k(const v8i* ops, float* out, int iters, unsigned long long* stat) {
  v8i a[4], b[4];
  for (int i = 0; i < 4; i++) { a[i] = ops[(threadIdx.x + 64 * i) % 2048]; b[i] = ops[(threadIdx.x + 64 * i + 777) % 2048]; }
  v16f acc[4];
  for (int j = 0; j < 4; j++) for (int i = 0; i < 16; i++) acc[j][i] = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[j], b[(j + it) & 3], acc[j], 0, 0, 0, 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0; for (int j = 0; j < 4; j++) for (int i = 0; i < 16; i++) s += acc[j][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { stat[0] = t1 - t0; stat[1] = r1 - r0; }
}
int main() {
  std::vector<int> h(2048 * 8);
  v8i* dops; float* out; unsigned long long* stat;
  CK(hipMalloc(&dops, 2048 * 32)); CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&stat, 16));
  for (int mode = 0; mode < 3; mode++) {
    srand(1);
    for (auto& x : h) {
      if (mode == 0) x = 0x38383838;                       // constant 1.0
      else if (mode == 1) { unsigned r = 0; for (int b = 0; b < 4; b++) { unsigned e = 0x28 + rand() % 0x20; r |= (e | ((rand() & 1) << 7)) << (8 * b); } x = (int)r; }  // random sign/exponent/mantissa, |x| in [2^-2, 2^2)
      else { unsigned r = 0; for (int b = 0; b < 4; b++) { unsigned e = rand() % 0x7e; r |= (e | ((rand() & 1) << 7)) << (8 * b); } x = (int)r; }  // full-range random bytes (no NaN)
    }
    CK(hipMemcpy(dops, h.data(), 2048 * 32, hipMemcpyHostToDevice));
    for (int th = 256; th <= 512; th += 256) {
      int iters = 40000;
      hipLaunchKernelGGL(k, 256, th, 0, 0, dops, out, 2000, stat); CK(hipDeviceSynchronize());
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, 256, th, 0, 0, dops, out, iters, stat);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long st[2]; CK(hipMemcpy(st, stat, 16, hipMemcpyDeviceToHost));
      double flops = (double)iters * 4 * (th / 64) * 256 * 2.0 * 32 * 32 * 64;
      printf("[mfma fp8 32x32x64] data=%s waves/SIMD=%d : %.0f TFLOP/s, in-kernel clock %.3f GHz, %.1f cycles/MFMA/wave\n",
             mode == 0 ? "constant" : mode == 1 ? "random(moderate)" : "random(full-range)", th / 256, flops / (ms * 1e-3) / 1e12,
             (double)st[0] / (double)st[1] * 0.1, (double)st[0] / ((double)iters * 4));
    }
  }
  return 0;
}
