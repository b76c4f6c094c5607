// Development probe: how v_mfma_scale_f32_32x32x64_f8f6f4 takes its E8M0 block scales (groundwork for a block-scaled V).
// A = B = 1.0 (e4m3 0x38) everywhere, so D[i][j] = sum over the two 32-wide K blocks of 32 * 2^(sa(i, blk) - 127) * 2^(sb(j, blk) - 127).
// Lane l of A holds row l & 31, K block l >> 5 (32 fp8 values); the scale operand is one VGPR per lane, op_sel picks the byte.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_scale_probe.hip -o /tmp/mfma_scale_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void probe(float* out, int mode) {
    const int lane = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; i++) { a[i] = 0x38383838; b[i] = 0x38383838; }
    v16f c;
    for (int i = 0; i < 16; i++) c[i] = 0.0f;
    int sa = 127, sb = 127;
    if (mode == 1) sa = 127 + (lane >> 5);                       // K block 1 of every row of A scaled by 2
    if (mode == 2) sa = 127 + ((lane & 31) == 3 ? 2 : 0);        // row 3 of A scaled by 4 (both K blocks)
    if (mode == 3) sb = 127 + ((lane & 31) == 5 ? 3 : 0);        // column 5 of B scaled by 8
    if (mode == 4) sa = (127 + 1) << 8 | 127;                    // byte 1 holds the scale 2, byte 0 holds 1: op_sel below picks byte 1
    if (mode == 4) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 1, sa, 0, sb);
    else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    // D layout of the 32x32 result: lane = column (l & 31), rows 8 * (r / 4) + (r % 4) + 4 * (l >> 5) for register r
    for (int r = 0; r < 16; r++) {
        const int row = 8 * (r >> 2) + (r & 3) + 4 * (lane >> 5), col = lane & 31;
        out[mode * 1024 + row * 32 + col] = c[r];
    }
}

int main() {
    float* d; hipMalloc(&d, 5 * 1024 * 4);
    float h[5 * 1024];
    for (int m = 0; m < 5; m++) probe<<<1, 64>>>(d, m);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* what[5] = {"no scales (expect 64 everywhere)", "A: K block 1 x2 (expect 32 + 64 = 96 everywhere)", "A: row 3 x4 (expect row 3 = 256, else 64)",
                           "B: column 5 x8 (expect column 5 = 512, else 64)", "A: scale byte 1 = x2 via op_sel 1 (expect 128 everywhere)"};
    for (int m = 0; m < 5; m++) {
        const float* x = h + m * 1024;
        printf("mode %d: %s\n   D[0][0] %.0f  D[3][0] %.0f  D[3][5] %.0f  D[0][5] %.0f  D[31][31] %.0f  D[4][7] %.0f\n", m, what[m], x[0], x[3 * 32], x[3 * 32 + 5], x[5], x[31 * 32 + 31], x[4 * 32 + 7]);
    }
    return 0;
}
