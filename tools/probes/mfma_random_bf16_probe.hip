// Probe 7: attainable v_mfma_f32_32x32x16_bf16 rate on constant vs random bf16 operands (DVFS), as probe 4 for fp8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ __launch_bounds__(512) void k(const v4i* ops, float* out, int iters, unsigned long long* stat) {
  bf8 a[4], b[4];
  for (int i = 0; i < 4; i++) {
    v4i x = ops[(threadIdx.x + 64 * i) % 2048], y = ops[(threadIdx.x + 64 * i + 777) % 2048];
    __builtin_memcpy(&a[i], &x, 16); __builtin_memcpy(&b[i], &y, 16);
  }
  v16f acc[4];
  for (int j = 0; j < 4; j++) for (int i = 0; i < 16; i++) acc[j][i] = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it += 4) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b[(j + u) & 3], acc[j], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0; for (int j = 0; j < 4; j++) for (int i = 0; i < 16; i++) s += acc[j][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { stat[0] = t1 - t0; stat[1] = r1 - r0; }
}
int main() {
  std::vector<int> h(2048 * 4);
  v4i* dops; float* out; unsigned long long* stat;
  CK(hipMalloc(&dops, 2048 * 16)); CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&stat, 16));
  for (int mode = 0; mode < 2; mode++) {
    srand(1);
    for (auto& x : h) {
      if (mode == 0) x = 0x3f803f80;  // 1.0, 1.0
      else { unsigned r = 0; for (int b = 0; b < 2; b++) { unsigned e = 0x3e00 + rand() % 0x300; r |= (e | ((rand() & 1) << 15)) << (16 * b); } x = (int)r; }  // random sign, |x| in [0.125, 8)
    }
    CK(hipMemcpy(dops, h.data(), 2048 * 16, hipMemcpyHostToDevice));
    for (int th = 256; th <= 512; th += 256) {
      int iters = 40000;
      hipLaunchKernelGGL(k, 256, th, 0, 0, dops, out, 2000, stat); CK(hipDeviceSynchronize());
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, 256, th, 0, 0, dops, out, iters, stat);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long st[2]; CK(hipMemcpy(st, stat, 16, hipMemcpyDeviceToHost));
      double flops = (double)iters * 4 * (th / 64) * 256 * 2.0 * 32 * 32 * 16;
      printf("[mfma bf16 32x32x16] data=%s waves/SIMD=%d : %.0f TFLOP/s, in-kernel clock %.3f GHz\n",
             mode == 0 ? "constant" : "random", th / 256, flops / (ms * 1e-3) / 1e12, (double)st[0] / (double)st[1] * 0.1);
    }
  }
  return 0;
}
