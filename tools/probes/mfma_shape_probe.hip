// Probe 5: does the clock the chip holds on RANDOM fp8 operands depend on the MFMA shape?  (MI355X_MICROARCH.md, DVFS
// give-back item 7, measured for bf16: 16x16x32 delivered ~1.15x the FLOP/s of 32x32x16 at equal cycles per FLOP.)
// Same FLOPs per wave in both kernels: 4 accumulators of 32x32 (x64) vs 8 accumulators of 16x16 (x128).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ __launch_bounds__(512) void k32(const v8i* ops, float* out, int iters, unsigned long long* stat) {
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
__global__ __launch_bounds__(512) void k16(const v8i* ops, float* out, int iters, unsigned long long* stat) {
  v8i a[4], b[4];
  for (int i = 0; i < 4; i++) { a[i] = ops[(threadIdx.x + 64 * i) % 2048]; b[i] = ops[(threadIdx.x + 64 * i + 777) % 2048]; }
  v4f acc[8];
  for (int j = 0; j < 8; j++) for (int i = 0; i < 4; i++) acc[j][i] = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[j & 3], b[(j + it) & 3], acc[j], 0, 0, 0, 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0; for (int j = 0; j < 8; j++) for (int i = 0; i < 4; i++) s += acc[j][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { stat[0] = t1 - t0; stat[1] = r1 - r0; }
}
int main() {
  std::vector<int> h(2048 * 8);
  v8i* dops; float* out; unsigned long long* stat;
  CK(hipMalloc(&dops, 2048 * 32)); CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&stat, 16));
  for (int mode = 0; mode < 2; mode++) {
    srand(1);
    for (auto& x : h) {
      if (mode == 0) x = 0x38383838;
      else { unsigned r = 0; for (int b = 0; b < 4; b++) { unsigned e = 0x28 + rand() % 0x20; r |= (e | ((rand() & 1) << 7)) << (8 * b); } x = (int)r; }
    }
    CK(hipMemcpy(dops, h.data(), 2048 * 32, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; rep++)
    for (int shape = 0; shape < 2; shape++)
    for (int th = 256; th <= 512; th += 256) {
      int iters = 40000;
      if (shape == 0) hipLaunchKernelGGL(k32, 256, th, 0, 0, dops, out, 2000, stat); else hipLaunchKernelGGL(k16, 256, th, 0, 0, dops, out, 2000, stat);
      CK(hipDeviceSynchronize());
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      if (shape == 0) hipLaunchKernelGGL(k32, 256, th, 0, 0, dops, out, iters, stat); else hipLaunchKernelGGL(k16, 256, th, 0, 0, dops, out, iters, stat);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long st[2]; CK(hipMemcpy(st, stat, 16, hipMemcpyDeviceToHost));
      double flops = (double)iters * 4 * (th / 64) * 256 * 2.0 * 32 * 32 * 64;
      printf("[mfma fp8 %s] data=%s waves/SIMD=%d rep=%d : %.0f TFLOP/s, in-kernel clock %.3f GHz, %.1f cycles per 131072 MACs per wave\n",
             shape == 0 ? "32x32x64 " : "16x16x128", mode == 0 ? "constant" : "random", th / 256, rep, flops / (ms * 1e-3) / 1e12,
             (double)st[0] / (double)st[1] * 0.1, (double)st[0] / ((double)iters * 4));
    }
  }
  return 0;
}
