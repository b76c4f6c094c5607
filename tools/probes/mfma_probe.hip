// Bring-up probe for gfx950 FP8 MFMA (32x32x64 f8f6f4): operand/accumulator lane maps,
// scaled vs unscaled semantics, issue rates, fp8 conversion behaviour.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe tools/probes/mfma_probe.hip
// Test infrastructure only (not part of the product path).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
#include <cstring>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef long v1l;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// decode e4m3fn / e5m2 on host
static float e4m3_to_f(uint8_t b) {
  int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 15 && m == 7) v = NAN;
  else if (e == 0) v = ldexpf((float)m, -9);
  else v = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -v : v;
}
static float e5m2_to_f(uint8_t b) {
  int s = b >> 7, e = (b >> 2) & 31, m = b & 3;
  float v;
  if (e == 31) v = m ? NAN : INFINITY;
  else if (e == 0) v = ldexpf((float)m, -16);
  else v = ldexpf(1.0f + m / 4.0f, e - 15);
  return s ? -v : v;
}

template <int MODE, int CBSZ, int BLGP>
__global__ void k_layout(const v8i* a, const v8i* b, v16f* c) {
  v16f acc = {0};
  v8i av = a[threadIdx.x], bv = b[threadIdx.x];
  if (MODE == 0) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, CBSZ, BLGP, 0, 0, 0, 0);
  if (MODE == 1) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, CBSZ, BLGP, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  if (MODE == 2) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, CBSZ, BLGP, 0, 0x80808080, 0, 0x7f7f7f7f);
  c[threadIdx.x] = acc;
}

// rate kernels: NACC independent accumulators, ITER iterations
template <int KIND>
__global__ __launch_bounds__(512) void k_rate(float* out, int iters, unsigned long long* cyc) {
  v8i a, b;
  for (int i = 0; i < 8; i++) { a[i] = 0x38383838 + threadIdx.x * (i + 1) * 0x01010101 % 7; b[i] = 0x30303030 + i; }
  v16f acc[4];
  for (int j = 0; j < 4; j++) for (int i = 0; i < 16; i++) acc[j][i] = 0.f;
  v4f acc4[4];
  for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) acc4[j][i] = 0.f;
  long a1 = 0x3838383838383838L + threadIdx.x, b1 = 0x3030303030303030L;
  v8s ah, bh;
  for (int i = 0; i < 8; i++) { ah[i] = 0x3f80 + threadIdx.x % 3; bh[i] = 0x3f00 + i; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (KIND == 0) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      if (KIND == 1) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], 0, 0, 0, 0, 0, 0);
      if (KIND == 2) acc4[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc4[j], 0, 0, 0, 0, 0, 0);
      if (KIND == 3) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a1, b1, acc[j], 0, 0, 0);
      if (KIND == 4) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
      if (KIND == 5) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], 1, 1, 0, 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int j = 0; j < 4; j++) { for (int i = 0; i < 16; i++) s += acc[j][i]; for (int i = 0; i < 4; i++) s += acc4[j][i]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// VALU rate: KIND 0 = v_exp_f32, 1 = v_fma_f32, 2 = cvt_pk_fp8, 3 = max3
template <int KIND>
__global__ __launch_bounds__(512) void k_valu(float* out, int iters, unsigned long long* cyc, float seed) {
  float x[16];
  for (int i = 0; i < 16; i++) x[i] = seed + threadIdx.x * 1e-3f + i;
  int r[8] = {0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (KIND == 0) x[i] = __builtin_amdgcn_exp2f(x[i]);
      if (KIND == 1) x[i] = __builtin_fmaf(x[i], seed, 0.5f);
      if (KIND == 3) x[i] = __builtin_fmaxf(__builtin_fmaxf(x[i], x[(i + 1) & 15]), x[(i + 2) & 15]);
    }
    if (KIND == 2) {
#pragma unroll
      for (int i = 0; i < 8; i++) { r[i] = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], r[i], false); }
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("" : "+v"(x[i]));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; i++) s += x[i];
  for (int i = 0; i < 8; i++) s += (float)r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void k_cvt(const float* x, unsigned* y8, unsigned* y5, int n) {
  int i = threadIdx.x;
  if (i < n) {
    int r = 0; r = __builtin_amdgcn_cvt_pk_fp8_f32(x[i], 0.f, r, false); y8[i] = r & 0xff;
    int q = 0; q = __builtin_amdgcn_cvt_pk_bf8_f32(x[i], 0.f, q, false); y5[i] = q & 0xff;
  }
}

template <int CBSZ, int BLGP>
static int layout_test(const char* name) {
  // A lane L=(h,row i) byte j ; B lane L=(h,col n) byte j ; hypothesis: C[i][n] = sum_{h,j} A[(h,i)][j]*B[(h,n)][j]
  std::vector<uint8_t> A(64 * 32), B(64 * 32);
  srand(123);
  auto rnd8 = [&](bool bf8) -> uint8_t {
    // small exact values: e4m3: {0,±1,±2,±0.5,±3}; e5m2 {0,±1,±2,±0.5,±3}
    int v = rand() % 9;
    static const float vals[9] = {0, 1, -1, 2, -2, 0.5f, -0.5f, 3, -3};
    float f = vals[v];
    for (int b = 0; b < 256; b++) { float d = bf8 ? e5m2_to_f(b) : e4m3_to_f(b); if (d == f && !(b == 0x80)) return (uint8_t)b; }
    return 0;
  };
  for (auto& x : A) x = rnd8(CBSZ == 1);
  for (auto& x : B) x = rnd8(BLGP == 1);
  v8i *da, *db; v16f* dc;
  CK(hipMalloc(&da, 2048)); CK(hipMalloc(&db, 2048)); CK(hipMalloc(&dc, 64 * 64));
  CK(hipMemcpy(da, A.data(), 2048, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, B.data(), 2048, hipMemcpyHostToDevice));
  std::vector<float> C(64 * 16);
  int bad_total = 0;
  for (int mode = 0; mode < 3; mode++) {
    if (mode == 0) hipLaunchKernelGGL((k_layout<0, CBSZ, BLGP>), 1, 64, 0, 0, da, db, dc);
    if (mode == 1) hipLaunchKernelGGL((k_layout<1, CBSZ, BLGP>), 1, 64, 0, 0, da, db, dc);
    if (mode == 2) hipLaunchKernelGGL((k_layout<2, CBSZ, BLGP>), 1, 64, 0, 0, da, db, dc);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(C.data(), dc, 64 * 64, hipMemcpyDeviceToHost));
    int bad = 0; double ratio = 0; int nr = 0;
    for (int lane = 0; lane < 64; lane++) for (int r = 0; r < 16; r++) {
      int n = lane & 31, i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      float ref = 0;
      for (int h = 0; h < 2; h++) for (int j = 0; j < 32; j++) {
        float av = CBSZ == 1 ? e5m2_to_f(A[(h * 32 + i) * 32 + j]) : e4m3_to_f(A[(h * 32 + i) * 32 + j]);
        float bv = BLGP == 1 ? e5m2_to_f(B[(h * 32 + n) * 32 + j]) : e4m3_to_f(B[(h * 32 + n) * 32 + j]);
        ref += av * bv;
      }
      float got = C[lane * 16 + r];
      if (got != ref) bad++;
      if (ref != 0) { ratio += got / ref; nr++; }
    }
    printf("[layout %s] mode=%d (0=scale0,1=scale0x7f,2=scaleA0x80) mismatches=%d/1024 mean(got/ref)=%.4f\n", name, mode, bad, nr ? ratio / nr : 0.0);
    if (mode < 2) bad_total += bad;
  }
  hipFree(da); hipFree(db); hipFree(dc);
  return bad_total;
}

template <int KIND>
static void rate_test(const char* name, double flop_per_mfma, int threads) {
  int blocks = 256 * (threads == 256 ? 1 : 1);
  float* out; unsigned long long* cyc;
  CK(hipMalloc(&out, blocks * threads * 4)); CK(hipMalloc(&cyc, 8));
  int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_rate<KIND>), blocks, threads, 0, 0, out, 100, cyc);
  CK(hipDeviceSynchronize());
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_rate<KIND>), blocks, threads, 0, 0, out, iters, cyc);
  hipEventRecord(e1);
  CK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  double n_mfma_wave = (double)iters * 4;
  double total = n_mfma_wave * (threads / 64) * blocks * flop_per_mfma;
  printf("[rate %-28s] waves/SIMD=%d  cycles/MFMA/wave=%.1f  chip=%.1f TFLOP/s  (%.2f ms, clk~%.2f GHz)\n", name, threads / 256,
         (double)c / n_mfma_wave, total / (ms * 1e-3) / 1e12, ms, (double)c / (ms * 1e-3) / 1e9 * 1.0);
  hipFree(out); hipFree(cyc);
}

template <int KIND>
static void valu_test(const char* name, int threads, int ops_per_iter) {
  int blocks = 256;
  float* out; unsigned long long* cyc;
  CK(hipMalloc(&out, blocks * threads * 4)); CK(hipMalloc(&cyc, 8));
  int iters = 20000;
  hipLaunchKernelGGL((k_valu<KIND>), blocks, threads, 0, 0, out, iters, cyc, 0.999f);
  CK(hipDeviceSynchronize());
  unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  printf("[valu %-12s] waves/SIMD=%d cycles/op/wave=%.2f\n", name, threads / 256, (double)c / ((double)iters * ops_per_iter));
  hipFree(out); hipFree(cyc);
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s arch=%s CUs=%d clock=%d kHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate);
  int bad = 0;
  bad += layout_test<0, 0>("e4m3 x e4m3");
  bad += layout_test<1, 1>("e5m2 x e5m2");
  bad += layout_test<1, 0>("A=e5m2 B=e4m3");
  printf("LAYOUT_HYPOTHESIS %s\n", bad == 0 ? "OK" : "FAILED");

  // cvt behaviour
  {
    float xs[] = {0.3f, 1.0f, 448.f, 464.f, 480.f, 500.f, 1e9f, -500.f, 0.0009765625f, 0.0029296875f, 0.001953125f * 1.5f, INFINITY, NAN, 57344.f, 65536.f, 1e-9f, 0.0625f*1.0625f, 0.0625f*1.1875f, 17.f, 18.f, 19.f, 27.f};
    int n = sizeof(xs) / 4;
    float* dx; unsigned *d8, *d5;
    CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&d8, n * 4)); CK(hipMalloc(&d5, n * 4));
    CK(hipMemcpy(dx, xs, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_cvt, 1, 64, 0, 0, dx, d8, d5, n);
    std::vector<unsigned> y8(n), y5(n);
    CK(hipMemcpy(y8.data(), d8, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(y5.data(), d5, n * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) printf("[cvt] x=%-14g e4m3=0x%02x (%g)  e5m2=0x%02x (%g)\n", xs[i], y8[i], e4m3_to_f(y8[i]), y5[i], e5m2_to_f(y5[i]));
  }

  const double F32 = 2.0 * 32 * 32 * 64, F16 = 2.0 * 16 * 16 * 128, FL = 2.0 * 32 * 32 * 16;
  for (int th = 256; th <= 512; th += 256) {
    rate_test<0>("scale(0x7f) 32x32x64 e4m3", F32, th);
    rate_test<1>("unscaled 32x32x64 e4m3", F32, th);
    rate_test<5>("unscaled 32x32x64 e5m2", F32, th);
    rate_test<2>("unscaled 16x16x128 e4m3", F16, th);
    rate_test<3>("legacy 32x32x16 fp8", FL, th);
    rate_test<4>("bf16 32x32x16", FL, th);
    valu_test<0>("v_exp_f32", th, 16);
    valu_test<1>("v_fma_f32", th, 16);
    valu_test<2>("cvt_pk_fp8", th, 8);
    valu_test<3>("2x v_max", th, 16);
  }
  return 0;
}
