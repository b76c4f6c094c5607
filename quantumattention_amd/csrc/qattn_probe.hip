// qattn_probe.hip -- MEASUREMENT entry of the C ABI (include/qattn.h: qattn_mfma_probe): what a bare
// v_mfma_f32_32x32x64_f8f6f4 loop sustains on THIS device, on the caller's (random) operand bytes, at the occupancy of the
// attention kernel (two waves per SIMD, one 512-thread workgroup per CU).  bench.py reports it as roofline.practical_peak next to
// the nominal 5 PFLOP/s: the chip lowers its clock under dense matrix work (MI355X_MICROARCH.md, DVFS give-back), so the
// attainable rate is a property of the device and the data, and the driver can re-observe it every round.  Nothing in the
// attention path calls this.
#include "qattn_common.h"

namespace {

typedef int v8i_ __attribute__((ext_vector_type(8)));
typedef float v16f_ __attribute__((ext_vector_type(16)));

constexpr int kProbeOperandVecs = 2048;                  // 32-byte operand vectors the caller fills with fp8 bytes
constexpr int kProbeThreads = 512;                       // 8 waves = two per SIMD, as the attention kernel
constexpr size_t kProbeOperandBytes = (size_t)kProbeOperandVecs * 32;

// every wave: 4 independent accumulators, A / B operands in registers (B rotated per iteration: nothing can be hoisted),
// iters x 4 MFMAs between two pairs of stamps {s_memtime (shader cycles), s_memrealtime (100 MHz)}
__global__ __launch_bounds__(kProbeThreads) void mfma_probe_kernel(const v8i_* ops, unsigned long long* stamps, float* sink, int iters) {
    v8i_ a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        a[i] = ops[(threadIdx.x + 64 * i + 131 * blockIdx.x) % kProbeOperandVecs];
        b[i] = ops[(threadIdx.x + 64 * i + 777 + 257 * blockIdx.x) % kProbeOperandVecs];
    }
    v16f_ acc[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[j][i] = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[j], b[(j + it) & 3], acc[j], QATTN_FMT_E4M3, QATTN_FMT_E4M3, 0, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 16; i++) s += acc[j][i];
    if (s == 1.2345e-30f) sink[0] = s;   // keeps the products alive; never true on real data
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (kProbeThreads / 64) + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

int probe_grid() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return n;
}

}  // namespace

extern "C" size_t qattn_mfma_probe_bytes(void) {
    const int grid = probe_grid();
    // [operands | {cycles, ticks} per wave | one sink word]
    return kProbeOperandBytes + sizeof(unsigned long long) * 2 * (size_t)(grid > 0 ? grid : 256) * (kProbeThreads / 64) + 16;
}

extern "C" int qattn_mfma_probe(void* scratch, size_t scratch_bytes, int iters, double* flops_per_launch, int* waves, void* stream) {
    if (!scratch || iters <= 0) return QATTN_ERR_INVALID_ARG;
    if (scratch_bytes < qattn_mfma_probe_bytes()) return QATTN_ERR_WORKSPACE;
    const int grid = probe_grid();
    if (grid <= 0) return QATTN_ERR_DEVICE;
    unsigned char* base = (unsigned char*)scratch;
    unsigned long long* stamps = (unsigned long long*)(base + kProbeOperandBytes);
    const int nw = grid * (kProbeThreads / 64);
    float* sink = (float*)(stamps + 2 * (size_t)nw);
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(grid), dim3(kProbeThreads), 0, (hipStream_t)stream, (const v8i_*)base, stamps, sink, iters);
    if (flops_per_launch) *flops_per_launch = (double)iters * 4.0 * nw * (2.0 * 32 * 32 * 64);
    if (waves) *waves = nw;
    return hipGetLastError() == hipSuccess ? QATTN_OK : QATTN_ERR_LAUNCH;
}
