// Probe 6: does the accumulator's register file matter for how much VALU work fits beside an MFMA?
// Slot = 1 v_mfma_f32_32x32x64_f8f6f4 (C/D in AccVGPRs "a" or in architectural VGPRs "v", or C = inline 0) + K v_fma_f32.
// Reports mean s_memtime cycles per slot per wave for 1 and 2 waves per SIMD.  Test infrastructure only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// FORM: 0 = C/D in VGPRs, 1 = C/D in AGPRs, 2 = D in VGPRs with C = 0 (no accumulator read)
template <int FORM, int K>
__global__ __launch_bounds__(512) void k_mix(float* out, int iters, float seed, unsigned long long* cyc) {
  v8i a, b;
  for (int i = 0; i < 8; i++) { a[i] = 0x38383838 + (threadIdx.x * (i + 1)) % 5; b[i] = 0x30303030 + i; }
  v16f acc[4];
  for (int j = 0; j < 4; j++) for (int i = 0; i < 16; i++) acc[j][i] = 0.f;
  float x[16];
  for (int i = 0; i < 16; i++) x[i] = seed + threadIdx.x * 1e-3f + i * 0.01f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (FORM == 0) asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a), "v"(b));
      if (FORM == 1) asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0" : "+a"(acc[u]) : "v"(a), "v"(b));
      if (FORM == 2) asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, 0" : "=v"(acc[u]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < K; j++) {
        const int i = (u * K + j) & 15;
        x[i] = __builtin_fmaf(x[i], seed, 0.25f);
        asm volatile("" : "+v"(x[i]));
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) atomicAdd(cyc, t1 - t0);
  float s = 0;
  for (int i = 0; i < 16; i++) s += x[i] + acc[0][i] + acc[1][i] + acc[2][i] + acc[3][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int FORM, int K>
static void run(const char* name) {
  for (int threads = 256; threads <= 512; threads += 256) {
    int blocks = 256; float* out; CK(hipMalloc(&out, blocks * threads * 4));
    unsigned long long* cyc; CK(hipMalloc(&cyc, 8));
    int iters = 4000;
    hipLaunchKernelGGL((k_mix<FORM, K>), blocks, threads, 0, 0, out, 200, 0.999f, cyc);
    CK(hipMemset(cyc, 0, 8)); CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mix<FORM, K>), blocks, threads, 0, 0, out, iters, 0.999f, cyc);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    double cps = (double)c / ((double)blocks * (threads / 64)) / ((double)iters * 4);
    double slots_per_s = (double)blocks * (threads / 64) * iters * 4 / (ms * 1e-3);
    printf("%-12s K=%2d waves/SIMD=%d : %.1f cycles/slot/wave, %.0f TFLOP/s MFMA, SIMD ns/slot %.2f\n", name, K, threads / 256, cps,
           slots_per_s * 2.0 * 32 * 32 * 64 / 1e12, 1e9 / (slots_per_s / 1024));
    CK(hipFree(out)); CK(hipFree(cyc));
  }
}
int main() {
  run<0, 0>("C/D=VGPR"); run<1, 0>("C/D=AGPR"); run<2, 0>("C=0,D=VGPR");
  run<0, 8>("C/D=VGPR"); run<1, 8>("C/D=AGPR"); run<2, 8>("C=0,D=VGPR");
  run<0, 12>("C/D=VGPR"); run<1, 12>("C/D=AGPR"); run<2, 12>("C=0,D=VGPR");
  run<0, 16>("C/D=VGPR"); run<1, 16>("C/D=AGPR"); run<2, 16>("C=0,D=VGPR");
  run<0, 24>("C/D=VGPR"); run<1, 24>("C/D=AGPR"); run<2, 24>("C=0,D=VGPR");
  return 0;
}
