// Probe 3: v_cvt_pk_u8_f32 rounding / saturation and issue rate (candidate for building e4m3 bytes of 2^x directly).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void k_sem(const float* x, unsigned* y, int n) {
  int i = threadIdx.x;
  if (i < n) y[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 1, 0xAABBCCDDu);
}
template <int KIND>
__global__ __launch_bounds__(512) void k_rate(float* out, int iters, float seed, unsigned long long* cyc) {
  float x[16]; unsigned r[4] = {0, 0, 0, 0};
  for (int i = 0; i < 16; i++) x[i] = seed * (threadIdx.x & 7) + i;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if (KIND == 0) r[i & 3] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], (i >> 2) & 3, r[i & 3]);
        if (KIND == 1) { r[i & 3] = __builtin_amdgcn_perm(r[i & 3], __float_as_uint(x[i]), 0x07060004u + i); }
        asm volatile("" : "+v"(x[i]));
      }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) atomicAdd(cyc, t1 - t0);
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(r[0] + r[1] + r[2] + r[3]) + x[3];
}
template <int KIND> static void rate(const char* name, int threads) {
  float* out; unsigned long long* cyc; CK(hipMalloc(&out, 256 * threads * 4)); CK(hipMalloc(&cyc, 8)); CK(hipMemset(cyc, 0, 8));
  int iters = 2000;
  hipLaunchKernelGGL((k_rate<KIND>), 256, threads, 0, 0, out, iters, 0.5f, cyc); CK(hipDeviceSynchronize());
  unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  printf("[rate %-14s] waves/SIMD=%d  %.2f cycles/op/wave\n", name, threads / 256, (double)c / (256.0 * threads / 64) / (iters * 64.0));
  CK(hipFree(out)); CK(hipFree(cyc));
}
int main() {
  float xs[] = {-5.f, -0.6f, -0.5f, -0.4f, 0.f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 3.5f, 126.4f, 126.5f, 254.5f, 255.f, 255.5f, 256.f, 1000.f, 1e9f, INFINITY, -INFINITY, NAN};
  int n = sizeof(xs) / 4; float* dx; unsigned* dy; CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dy, n * 4));
  CK(hipMemcpy(dx, xs, n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_sem, 1, 64, 0, 0, dx, dy, n); unsigned ys[64]; CK(hipMemcpy(ys, dy, n * 4, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) printf("[cvt_pk_u8] x=%-10g -> dword 0x%08x (byte1=%u)\n", xs[i], ys[i], (ys[i] >> 8) & 255);
  rate<0>("cvt_pk_u8_f32", 256); rate<0>("cvt_pk_u8_f32", 512); rate<1>("v_perm_b32", 256); rate<1>("v_perm_b32", 512);
  return 0;
}
