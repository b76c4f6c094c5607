// Probe 5: v_cvt_pknorm_u16_f32 semantics (rounding, clamping, NaN/inf) and the issue rates of v_pk_fma_f32,
// v_cvt_pknorm_u16_f32 and v_perm_b32 -- candidates for a cheaper e4m3-byte exponential (5 VALU per 4 scores).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__global__ void k_sem(const float* x, unsigned* y, int n) {
  int i = threadIdx.x;
  if (i < n) { us2 r = __builtin_amdgcn_cvt_pknorm_u16(x[i], x[i] * 0.5f); unsigned u; __builtin_memcpy(&u, &r, 4); y[i] = u; }
}
template <int KIND>
__global__ __launch_bounds__(512) void k_rate(float* out, int iters, float seed, unsigned long long* cyc) {
  f2 x[8]; unsigned r[4] = {0, 0, 0, 0}; f2 a = {seed, seed}, b = {1.0f, 2.0f};
  for (int i = 0; i < 8; i++) x[i] = f2{seed * (threadIdx.x & 7) + i, seed + i};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (KIND == 0) { x[i] = __builtin_elementwise_fma(x[i], a, b); }
        if (KIND == 1) { us2 q = __builtin_amdgcn_cvt_pknorm_u16(x[i][0], x[i][1]); unsigned uu; __builtin_memcpy(&uu, &q, 4); r[i & 3] ^= uu; }
        if (KIND == 2) { x[i][0] = __builtin_fmaf(x[i][0], seed, 1.0f); }
        asm volatile("" : "+v"(x[i]));
      }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) atomicAdd(cyc, t1 - t0);
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(r[0] + r[1] + r[2] + r[3]) + x[3][0] + x[5][1];
}
template <int KIND> static void rate(const char* name, int threads) {
  float* out; unsigned long long* cyc; CK(hipMalloc(&out, 256 * threads * 4)); CK(hipMalloc(&cyc, 8)); CK(hipMemset(cyc, 0, 8));
  int iters = 2000;
  hipLaunchKernelGGL((k_rate<KIND>), 256, threads, 0, 0, out, iters, 0.5f, cyc); CK(hipDeviceSynchronize());
  unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  printf("[rate %-22s] waves/SIMD=%d  %.2f cycles/instr/wave (incl. 1 xor for pknorm)\n", name, threads / 256, (double)c / (256.0 * threads / 64) / (iters * 64.0));
  CK(hipFree(out)); CK(hipFree(cyc));
}
int main() {
  const float u = 1.0f / 65535.0f;
  float xs[] = {-1.f, -u, 0.f, 0.4f * u, 0.5f * u, 0.6f * u, 1.5f * u, 2.5f * u, 3.5f * u, 120.f * u, 120.5f * u, 121.5f * u, 255.f * u, 256.f * u, 0.5f, 1.0f, 2.0f, 1e9f, INFINITY, -INFINITY, NAN};
  int n = sizeof(xs) / 4; float* dx; unsigned* dy; CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dy, n * 4));
  CK(hipMemcpy(dx, xs, n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_sem, 1, 64, 0, 0, dx, dy, n); unsigned ys[64]; CK(hipMemcpy(ys, dy, n * 4, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) printf("[cvt_pknorm_u16] x*65535=%-12g -> lo=%u hi(x/2)=%u\n", xs[i] * 65535.0, ys[i] & 0xffff, ys[i] >> 16);
  rate<0>("v_pk_fma_f32", 256); rate<0>("v_pk_fma_f32", 512);
  rate<1>("v_cvt_pknorm_u16_f32", 256); rate<1>("v_cvt_pknorm_u16_f32", 512);
  rate<2>("v_fma_f32", 256); rate<2>("v_fma_f32", 512);
  return 0;
}
