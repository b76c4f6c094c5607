// energy_probe.hip -- what each instruction CLASS of the C2 attention kernel costs at the chip's power cap (round 6).
//
// The fused kernel sits at the 1400 W cap: a removed stall comes back as a lower clock, only removed ENERGY shortens a launch
// (DESIGN.md section 4.3).  At the cap, time IS energy (J = 1400 W x s), so the marginal launch time of adding one class of work to a
// bare-MFMA loop -- at the kernel's occupancy (one 512-thread workgroup per CU, two waves per SIMD), on random operand bytes -- is that
// class's share of the energy budget.  One "iteration" of a wave mimics one 64-key iteration of attn_fwd_kernel_v2 (D = 128):
//     8 x v_mfma_f32_32x32x64_f8f6f4 (QK^T 4 + PV 4)  [+ NSUM x v_mfma_f32_16x16x128_f8f6f4: row sums (1) and N_eff (2)]
//     NV VALU instructions in the kernel's mix (per 32 scores: 16 v_pk_fma_f32, 16 v_cvt_pknorm_u16_f32, 8 v_perm_b32, 12 v_max3_f32,
//        32 v_mov_b32 (accumulator zeroing), the rest v_add_f32 / v_mul_f32)
//     NL ds_read_b128 (operand fragments: conflict-free 16-byte reads), ND global_load_lds_dwordx4 (K/V ring: 1 KiB per wave each, L2-resident source)
//     a workgroup barrier + vmcnt(0) every second (EVERY-th) iteration when BAR
// Test infrastructure only; nothing here is linked into the product.  Build + run on the GPU box:
//     hipcc --offload-arch=gfx950 -O3 tools/probes/energy_probe.hip -o /tmp/energy_probe && /tmp/energy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kThreads = 512;
constexpr int kLds = 5 * 2 * 64 * 128;   // the kernel's K/V ring: 80 KiB

// one VALU instruction of the kernel's mix, on the wave's private registers (x: 32 floats, r: 8 dwords)
template <int I>
__device__ __forceinline__ void valu_one(float (&x)[32], unsigned (&r)[8], float c8, float off8) {
    constexpr int k = I % 83;    // position within one iteration's mix (83 explicit instructions + ~28 operand moves the compiler adds for the
                                 // v_pk_fma pairs = the kernel's 111 per iteration: counted in the probe's ISA)
    if (k < 16) {          // v_pk_fma_f32: two scores
        f2 a = {x[(2 * k) & 31], x[(2 * k + 1) & 31]}, cc = {c8, c8}, oo = {off8, off8}, d;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(cc), "v"(oo));
        x[(2 * k) & 31] = d[0]; x[(2 * k + 1) & 31] = d[1];
    } else if (k < 32) {   // v_cvt_pknorm_u16_f32
        unsigned d;
        asm volatile("v_cvt_pknorm_u16_f32 %0, %1, %2" : "=v"(d) : "v"(x[(2 * k) & 31]), "v"(x[(2 * k + 1) & 31]));
        r[k & 7] = d;
    } else if (k < 40) {   // v_perm_b32
        unsigned d;
        asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(r[k & 7]), "v"(r[(k + 1) & 7]), "v"(0x06040200u));
        r[(k + 2) & 7] = d;
    } else if (k < 52) {   // v_max3_f32
        float d;
        asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x[k & 31]), "v"(x[(k + 5) & 31]), "v"(x[(k + 9) & 31]));
        x[(k + 13) & 31] = d;
    } else if (k < 56) {   // v_mov_b32 (the rest of the kernel's moves come from the compiler: see above)
        float d;
        asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(x[(k + 3) & 31]));
        x[k & 31] = d;
    } else {               // v_add_f32 / v_mul_f32
        float d;
        if (k & 1) asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(x[k & 31]), "v"(x[(k + 7) & 31]));
        else asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(x[k & 31]), "v"(c8));
        x[(k + 11) & 31] = d;
    }
}
template <int I0, int N>
__device__ __forceinline__ void valu_block(float (&x)[32], unsigned (&r)[8], float c8, float off8) {
    if constexpr (N > 0) {
        valu_one<I0>(x, r, c8, off8);
        valu_block<I0 + 1, N - 1>(x, r, c8, off8);
    }
}

template <int NSUM, int NV, int NL, int ND, bool BAR, bool MFMA, int EVERY = 2>
__global__ __launch_bounds__(kThreads) void k_energy(const v8i* ops, const unsigned char* kv, float* sink, int iters, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v8i a[4], b[2], ones;
#pragma unroll
    for (int i = 0; i < 4; i++) a[i] = ops[(threadIdx.x + 64 * i + 131 * blockIdx.x) % 2048];
#pragma unroll
    for (int i = 0; i < 2; i++) b[i] = ops[(threadIdx.x + 64 * i + 777 + 257 * blockIdx.x) % 2048];
    ones = ops[(threadIdx.x + 999) % 2048];
    // LDS image: random bytes (the ring holds K / V fragments)
    for (int i = threadIdx.x; i < kLds / 16; i += kThreads) reinterpret_cast<v4i*>(smem)[i] = reinterpret_cast<const v4i*>(ops)[i % 4096];
    __syncthreads();
    v16f acc[4];
    v4f s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[j][i] = 0.0f;
    float x[32];
    unsigned r[8];
#pragma unroll
    for (int i = 0; i < 32; i++) x[i] = __int_as_float(0x3f000000 + ((a[i & 3][i & 7] >> 3) & 0x7fffff));   // random mantissas in [0.5, 1)
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = (unsigned)b[i & 1][i];
    const float c8 = 0.9990234f, off8 = 0.0004882f;
    const unsigned lds_lane = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + ((lane >> 5) << 10) + ((lane & 31) << 4);
    const unsigned char* src = kv + ((size_t)(blockIdx.x & 7) << 20) + (wave << 10) + (lane << 4);
    unsigned slot = 0, goff = 0;
    v4i frag[2];
    frag[0] = v4i{0, 0, 0, 0}; frag[1] = frag[0];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if (BAR && (it & (EVERY - 1)) == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (ND > 0) {
#pragma unroll
            for (int d = 0; d < ND; d++) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + goff + d * 8192),
                                                 (__attribute__((address_space(3))) void*)(smem + slot + d * 8192 + (wave << 10)), 16, 0, 0);
            }
            goff = (goff + 16384) & ((1u << 20) - 1);
            slot = slot + 16384 == (unsigned)kLds ? 0u : slot + 16384;
        }
        constexpr int PER = 8;   // MFMA slots per iteration
#pragma unroll
        for (int u = 0; u < PER; u++) {
            if (MFMA) acc[u & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[u & 3], b[(u + it) & 1], acc[u & 3], 0, 0, 0, 0, 0, 0);
            // this slot's share of the LDS reads and the VALU work
            constexpr int L0 = NL / PER, LX = NL % PER;
#pragma unroll
            for (int l = 0; l < L0 + 1; l++) {
                if (l < L0 || u < LX) {
                    v4i d;
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(lds_lane + ((unsigned)(u & 3) << 11) + ((unsigned)(it & 3) << 14)), "n"(512 * (l & 1)) : "memory");
                    frag[l & 1] = d;
                }
            }
            if (u == 0) valu_block<0, NV / PER + ((0 < NV % PER) ? 1 : 0)>(x, r, c8, off8);
            if (u == 1) valu_block<11, NV / PER + ((1 < NV % PER) ? 1 : 0)>(x, r, c8, off8);
            if (u == 2) valu_block<22, NV / PER + ((2 < NV % PER) ? 1 : 0)>(x, r, c8, off8);
            if (u == 3) valu_block<33, NV / PER + ((3 < NV % PER) ? 1 : 0)>(x, r, c8, off8);
            if (u == 4) valu_block<44, NV / PER + ((4 < NV % PER) ? 1 : 0)>(x, r, c8, off8);
            if (u == 5) valu_block<55, NV / PER + ((5 < NV % PER) ? 1 : 0)>(x, r, c8, off8);
            if (u == 6) valu_block<66, NV / PER + ((6 < NV % PER) ? 1 : 0)>(x, r, c8, off8);
            if (u == 7) valu_block<77, NV / PER + ((7 < NV % PER) ? 1 : 0)>(x, r, c8, off8);
            if (u == 3 && NSUM >= 1 && MFMA) s1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, b[it & 1], s1, 0, 0, 0, 0, 0, 0);
            if (u == 4 && NSUM >= 2 && MFMA) s2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, b[it & 1], s2, 0, 1, 0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (NL > 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // fold the fragments into an operand so that the reads are live (and the MFMA operands keep toggling)
            a[it & 3][0] ^= frag[0][0] & 0x07070707;
            a[(it + 1) & 3][4] ^= frag[1][1] & 0x07070707;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { stamps[2 * (blockIdx.x * 8 + wave)] = t1 - t0; stamps[2 * (blockIdx.x * 8 + wave) + 1] = r1 - r0; }
    float s = s1[0] + s2[0];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 16; i++) s += acc[j][i];
#pragma unroll
    for (int i = 0; i < 32; i++) s += x[i];
#pragma unroll
    for (int i = 0; i < 8; i++) s += (float)r[i];
    if (s == 1.2345e-30f) sink[0] = s;
}

struct Ctx { v8i* ops; unsigned char* kv; float* sink; int grid; unsigned long long* stamps; };

template <int NSUM, int NV, int NL, int ND, bool BAR, bool MFMA, int EVERY = 2>
static double run(const Ctx& c, const char* name, double base_ns) {
    auto kern = k_energy<NSUM, NV, NL, ND, BAR, MFMA, EVERY>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    const int iters = 6000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // settle at the power cap: 0.4 s of back-to-back launches, then the median of the following ones
    hipLaunchKernelGGL(kern, dim3(c.grid), dim3(kThreads), kLds, 0, c.ops, c.kv, c.sink, 200, c.stamps);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(c.grid), dim3(kThreads), kLds, 0, c.ops, c.kv, c.sink, iters, c.stamps);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms1; CK(hipEventElapsedTime(&ms1, e0, e1));
    const int warm = std::max(3, (int)(400.0 / ms1)), laps = std::max(5, (int)(300.0 / ms1));
    for (int i = 0; i < warm; i++) hipLaunchKernelGGL(kern, dim3(c.grid), dim3(kThreads), kLds, 0, c.ops, c.kv, c.sink, iters, c.stamps);
    std::vector<float> t;
    for (int i = 0; i < laps; i++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(c.grid), dim3(kThreads), kLds, 0, c.ops, c.kv, c.sink, iters, c.stamps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    const double ns = (double)t[t.size() / 2] * 1e6 / iters;   // ns per iteration (all waves of the chip advance one iteration)
    // the clock the chip held inside the last launch: shader cycles / 100 MHz ticks of every wave's loop (median)
    std::vector<unsigned long long> st(2 * (size_t)c.grid * 8);
    CK(hipMemcpy(st.data(), c.stamps, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> clk;
    for (size_t w = 0; w < st.size() / 2; w++) if (st[2 * w + 1]) clk.push_back((double)st[2 * w] / (double)st[2 * w + 1] * 0.1);
    std::sort(clk.begin(), clk.end());
    const double ghz = clk.empty() ? 0.0 : clk[clk.size() / 2];
    const double tf = MFMA ? 8.0 * 2 * 32 * 32 * 64 * (c.grid * 8.0) / (ns * 1e-9) / 1e12 : 0.0;
    printf("%-46s NSUM %d NV(explicit) %3d NL %2d ND %d BAR %d | %7.1f ns / iteration = %6.0f cycles at %.3f GHz | %6.0f TFLOP/s (8 products) | %+7.1f ns vs MFMA8+2 (%+5.1f %%)\n", name, NSUM, NV, NL, ND,
           (int)BAR, ns, ns * ghz, ghz, tf, base_ns > 0 ? ns - base_ns : 0.0, base_ns > 0 ? 100.0 * (ns - base_ns) / base_ns : 0.0);
    fflush(stdout);
    return ns;
}

int main() {
    Ctx c;
    int dev = 0; CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&c.grid, hipDeviceAttributeMultiprocessorCount, dev));
    std::vector<unsigned char> h(2048 * 32 * 2);
    unsigned s = 12345u;
    for (auto& b : h) { s = s * 1664525u + 1013904223u; const unsigned e = 0x28 + ((s >> 9) % 0x20); b = (unsigned char)(e | ((s >> 30) << 7)); }   // |x| in [2^-2, 2^2)
    CK(hipMalloc(&c.ops, h.size())); CK(hipMemcpy(c.ops, h.data(), h.size(), hipMemcpyHostToDevice));
    std::vector<unsigned char> kvh(8u << 20);
    for (auto& b : kvh) { s = s * 1664525u + 1013904223u; b = (unsigned char)(s >> 24); }
    CK(hipMalloc(&c.kv, kvh.size() + (1u << 16))); CK(hipMemcpy(c.kv, kvh.data(), kvh.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&c.sink, 64));
    CK(hipMalloc(&c.stamps, 16 * 8 * 1024));
    printf("energy_probe: %d workgroups x %d threads (two waves per SIMD), random e4m3 operands; ns per iteration = launch time / iterations at the settled clock\n", c.grid, kThreads);
    const double m8 = run<0, 0, 0, 0, false, true>(c, "8 products only (bare MFMA)", 0);
    run<1, 0, 0, 0, false, true>(c, "+ row-sum MFMA (FAST's matrix work)", m8);
    const double base = run<2, 0, 0, 0, false, true>(c, "+ row-sum + N_eff MFMA (AUTO's matrix work)", m8);
    run<2, 0, 16, 0, false, true>(c, "AUTO matrix + 16 ds_read_b128", base);
    run<2, 0, 32, 0, false, true>(c, "AUTO matrix + 32 ds_read_b128", base);
    run<2, 42, 0, 0, false, true>(c, "AUTO matrix + half the VALU", base);
    run<2, 83, 0, 0, false, true>(c, "AUTO matrix + 111 VALU (the kernel's count)", base);
    run<2, 0, 0, 2, true, true>(c, "AUTO matrix + 2 LDS-DMA + barrier / 2", base);
    run<2, 0, 0, 0, true, true>(c, "AUTO matrix + barrier / 2 only", base);
    run<2, 83, 16, 2, true, true>(c, "everything (one kernel iteration's mix)", base);
    run<1, 75, 16, 2, true, true>(c, "everything, FAST's mix (no N_eff, ~100 VALU)", base);
    run<2, 62, 16, 2, true, true>(c, "everything with -25 % VALU", base);
    run<2, 83, 8, 2, true, true>(c, "everything with 8 ds_read_b128 (-50 %)", base);
    run<2, 56, 18, 2, true, true>(c, "the kernel's ISA counts: ~90 VALU, 18 reads", base);
    run<1, 50, 18, 2, true, true>(c, "FAST's ISA counts: ~82 VALU, 18 reads, no N_eff", base);
    run<2, 56, 18, 2, true, true, 4>(c, "the kernel's ISA counts, barrier every 4 iterations", base);
    run<2, 56, 18, 2, true, true, 1>(c, "the kernel's ISA counts, barrier every iteration", base);
    run<2, 56, 18, 2, false, true>(c, "the kernel's ISA counts, no barrier", base);
    run<0, 56, 18, 2, true, false>(c, "everything BUT the matrix work (~90 VALU)", 0);
    run<0, 83, 0, 0, false, false>(c, "111 VALU alone (no matrix work)", 0);
    run<0, 0, 16, 0, false, false>(c, "16 ds_read_b128 alone", 0);
    return 0;
}
