// qattn_api.hip -- C-ABI entry points of libqattn_hip.so that are not in qattn_quant.hip (include/qattn.h).
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <vector>

#include "qattn_attn.h"

using namespace qattn;

// Development switch for A/B runs: QATTN_KERNEL_VARIANT = 1 (first, non-pipelined structure), 2 (default for D = 128:
// 8 waves x 32 rows, pipelined), 3 (experimental: 4 waves x 64 rows; correct but register-allocation bound).
// QATTN_EXACT_EXP=1 disables the byte-exponential fast path (see qattn_attn_v2.hip).
static int exact_exp() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("QATTN_EXACT_EXP");
        v = e ? atoi(e) : 0;
    }
    return v;
}

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

static int kernel_variant() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("QATTN_KERNEL_VARIANT");
        v = e ? atoi(e) : 2;
    }
    return v;
}

static hipEvent_t g_ev[2] = {nullptr, nullptr};

// development (QATTN_STEP_EVENTS=1): milliseconds between the HIP events recorded right before and right after the most
// recent attention launch(es) on their stream; synchronises on the second event.  Used by bench.py to time the attention
// kernel inside the fused step without a profiler.
extern "C" float qattn_debug_last_attention_ms(void) {
    if (!g_ev[0]) return -1.0f;
    float ms = -1.0f;
    if (hipEventSynchronize(g_ev[1]) != hipSuccess || hipEventElapsedTime(&ms, g_ev[0], g_ev[1]) != hipSuccess) return -1.0f;
    return ms;
}

// q16 != nullptr: the fused step -- Q is the bf16 tensor, quantised inside the kernel from q_amax_bits; scale_q is an OUTPUT.
static int attention_impl(const void* q8, const void* k8, const void* v8, void* out, float* lse, const float* scale_q,
                          const float* scale_k, const float* scale_v, int B, int Hq, int Hkv, int Sq, int Skv, int D, int qk_fmt,
                          int v_fmt, int out_fmt, int scale_mode, int is_causal, float sm_scale, const void* q16,
                          const unsigned* q_amax_bits, float* sq_out, int q_numerics, void* stream) {
    // argument checks mirror the reference launcher's TORCH_CHECKs (tk/attention.py:362-415)
    if ((!q8 && !q16) || !k8 || !v8 || !out || (!scale_q && !q16) || !scale_k) return QATTN_ERR_INVALID_ARG;
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || Sq <= 0 || Skv <= 0) return QATTN_ERR_INVALID_ARG;
    if (D != 64 && D != 128 && D != 256) return QATTN_ERR_UNSUPPORTED_DIM;  // nn.py:45-49
    if (Hq % Hkv != 0) return QATTN_ERR_UNSUPPORTED_DIM;                    // tk/attention.py:398-399
    if (qk_fmt != QATTN_FMT_E4M3 && qk_fmt != QATTN_FMT_E5M2) return QATTN_ERR_UNSUPPORTED_FMT;
    if (v_fmt != qk_fmt) return QATTN_ERR_UNSUPPORTED_FMT;
    if (out_fmt != QATTN_FMT_BF16 && out_fmt != QATTN_FMT_FP16) return QATTN_ERR_UNSUPPORTED_FMT;
    if (scale_mode != QATTN_SCALE_HEAD && scale_mode != QATTN_SCALE_TOKEN) return QATTN_ERR_INVALID_ARG;
    AttnParams p;
    p.q = (const unsigned char*)q8; p.k = (const unsigned char*)k8; p.v = (const unsigned char*)v8;
    p.out = out; p.lse = lse; p.sq = scale_q; p.sk = scale_k; p.sv = scale_v;
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.Sq = Sq; p.Skv = Skv;
    const bool use_v4_full = D != 128;  // D = 64 / 256: the templated three-waves kernel covers every case
    const bool use_v3 = !use_v4_full && kernel_variant() == 3;
    p.waves = (use_v3 || use_v4_full) ? kWaves : env_int("QATTN_V2_WAVES", 8);  // v3 workgroups also cover 256 rows
    p.lds_pad = env_int("QATTN_V2_LDS", 0);
    p.dbg = env_int("QATTN_V2_DBG", 0);
    p.dbg_buf = nullptr;
    static unsigned long long* dbg_dev = nullptr;
    const long n_dbg_waves = (long)B * Hq * ceil_div(Sq, p.waves * kQPerWave) * (use_v3 ? 4 : p.waves);
    if (p.dbg & 16) {
        if (!dbg_dev) (void)hipMalloc(&dbg_dev, sizeof(unsigned long long) * 2 * (1 << 20));
        p.dbg_buf = dbg_dev;
        (void)hipMemsetAsync(dbg_dev, 0, sizeof(unsigned long long) * 2 * n_dbg_waves, (hipStream_t)stream);
    }
    p.nqb = ceil_div(Sq, p.waves * kQPerWave);
    p.nchunks = ceil_div(Skv, 64);
    p.out_fmt = out_fmt;
    p.xcd_remap = ((B * Hq) % 8 == 0) ? 1 : 0;
    const float sm = sm_scale > 0.0f ? sm_scale : 1.0f / sqrtf((float)D);
    p.sm_log2e = sm * 1.4426950408889634f;
    p.exact_exp = exact_exp();
    p.use_v4 = kernel_variant() == 4 ? 1 : 0;
    p.q16 = (const unsigned char*)q16; p.q_amax_bits = q_amax_bits; p.sq_out = sq_out; p.q_numerics = q_numerics;
    hipStream_t st = (hipStream_t)stream;
    static const bool step_events = env_int("QATTN_STEP_EVENTS", 0) != 0;  // development: time this launch inside a longer sequence
    if (step_events) {
        if (!g_ev[0]) { (void)hipEventCreate(&g_ev[0]); (void)hipEventCreate(&g_ev[1]); }
        (void)hipEventRecord(g_ev[0], st);
    }
    int rc;
    if (use_v4_full) rc = launch_attn_v4_full(p, D, qk_fmt, is_causal, scale_mode, st);
    else if (use_v3) rc = launch_attn_v3(p, D, qk_fmt, is_causal, scale_mode, st);
    else rc = launch_attn_v2(p, D, qk_fmt, is_causal, scale_mode, st);
    if (step_events) (void)hipEventRecord(g_ev[1], st);
    if (rc != QATTN_OK) return rc;
    if ((p.dbg & 16) && p.dbg_buf) {  // diagnostic build path only: synchronises and prints per-wave sweep statistics
        static int printed = 0;
        (void)hipStreamSynchronize(st);
        if (printed++ == 3) {
            std::vector<unsigned long long> h(2 * n_dbg_waves);
            (void)hipMemcpy(h.data(), p.dbg_buf, sizeof(unsigned long long) * 2 * n_dbg_waves, hipMemcpyDeviceToHost);
            std::vector<double> cyc, clk;
            for (long i = 0; i < n_dbg_waves; i++) if (h[2 * i + 1]) { cyc.push_back((double)h[2 * i]); clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1); }
            if (!cyc.empty()) {
                std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
                const int iters = ceil_div(Skv, 64) + 2;
                if (p.dbg & 32) {
                    std::vector<unsigned long long> sg(64 * 8);
                    (void)hipMemcpy(sg.data(), p.dbg_buf + 2 * (1 << 19), sizeof(unsigned long long) * 64 * 8, hipMemcpyDeviceToHost);
                    double tot[6] = {0, 0, 0, 0, 0, 0};
                    for (int w = 0; w < 64; w++) for (int i = 0; i < 6; i++) tot[i] += (double)sg[w * 8 + i] / 64.0;
                    fprintf(stderr, "[qattn dbg] per-iteration segment cycles (mean of 64 waves): seg0 %.0f | seg1 %.0f | seg2 %.0f | seg3 %.0f | seg4 %.0f | seg5 %.0f\n",
                            tot[0] / iters, tot[1] / iters, tot[2] / iters, tot[3] / iters, tot[4] / iters, tot[5] / iters);
                    for (int w = 0; w < 8; w++)
                        fprintf(stderr, "[qattn dbg]   wave %d: %.0f %.0f %.0f %.0f %.0f %.0f\n", w, (double)sg[w * 8] / iters, (double)sg[w * 8 + 1] / iters,
                                (double)sg[w * 8 + 2] / iters, (double)sg[w * 8 + 3] / iters, (double)sg[w * 8 + 4] / iters, (double)sg[w * 8 + 5] / iters);
                }
                fprintf(stderr, "[qattn dbg] waves=%zu sweep cycles median=%.0f (%.1f per iteration over %d) p10=%.0f p90=%.0f | in-kernel clock median %.3f GHz\n",
                        cyc.size(), cyc[cyc.size() / 2], cyc[cyc.size() / 2] / iters, iters, cyc[cyc.size() / 10], cyc[cyc.size() * 9 / 10], clk[clk.size() / 2]);
            }
        }
    }
    return hipGetLastError() == hipSuccess ? QATTN_OK : QATTN_ERR_LAUNCH;
}

extern "C" int qattn_abi_version(void) { return QATTN_ABI_VERSION; }

extern "C" const char* qattn_strerror(int code) {
    switch (code) {
        case QATTN_OK: return "ok";
        case QATTN_ERR_INVALID_ARG: return "invalid argument (null pointer, non-positive dimension or unknown enum)";
        case QATTN_ERR_UNSUPPORTED_DIM: return "unsupported head dimension (need 64, 128 or 256) or Hq not divisible by Hkv";
        case QATTN_ERR_UNSUPPORTED_FMT: return "unsupported element format / layout combination";
        case QATTN_ERR_WORKSPACE: return "workspace missing or too small";
        case QATTN_ERR_LAUNCH: return "HIP kernel launch failed";
        case QATTN_ERR_DEVICE: return "current HIP device is not gfx950 (MI355X)";
        default: return "unknown qattn error code";
    }
}

extern "C" int qattn_check_device(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return QATTN_ERR_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return QATTN_ERR_DEVICE;
    return strstr(prop.gcnArchName, "gfx950") ? QATTN_OK : QATTN_ERR_DEVICE;
}

extern "C" int qattn_fp8_attention_forward(const void* q8, const void* k8, const void* v8, void* out, float* lse,
                                           const float* scale_q, const float* scale_k, const float* scale_v, int B,
                                           int Hq, int Hkv, int Sq, int Skv, int D, int qk_fmt, int v_fmt, int out_fmt,
                                           int scale_mode, int is_causal, float sm_scale, void* stream) {
    return attention_impl(q8, k8, v8, out, lse, scale_q, scale_k, scale_v, B, Hq, Hkv, Sq, Skv, D, qk_fmt, v_fmt, out_fmt,
                          scale_mode, is_causal, sm_scale, nullptr, nullptr, nullptr, 0, stream);
}

// true when the attention kernel can quantise Q itself (hand-scheduled D = 128 kernel, 8 waves, head-wise scales, bf16
// inputs, byte-exponential path): then the pre-pass skips Q's payload (one read and one write of Q less).
static bool q_fusion_ok(int D, int in_fmt, int scale_mode) {
    return D == 128 && in_fmt == QATTN_FMT_BF16 && scale_mode == QATTN_SCALE_HEAD && kernel_variant() == 2 && !exact_exp() &&
           env_int("QATTN_V2_WAVES", 8) == 8 && env_int("QATTN_NO_Q_FUSION", 0) == 0;
}

extern "C" int qattn_fp8_quant_attention_forward(const void* q, const void* k, const void* v, int in_fmt, void* out, void* q8,
                                                 void* k8, void* v8, float* scale_q, float* scale_k, float* scale_v, int B,
                                                 int Hq, int Hkv, int Sq, int Skv, int D, int fp8_fmt, int scale_mode,
                                                 int numerics, int is_causal, float sm_scale, void* workspace,
                                                 size_t workspace_bytes, void* stream) {
    if (!q || !k || !v || !out || !q8 || !k8 || !v8 || !scale_q || !scale_k || !scale_v) return QATTN_ERR_INVALID_ARG;
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || Sq <= 0 || Skv <= 0) return QATTN_ERR_INVALID_ARG;
    if (D != 64 && D != 128 && D != 256) return QATTN_ERR_UNSUPPORTED_DIM;
    if (Hq % Hkv != 0) return QATTN_ERR_UNSUPPORTED_DIM;
    if (scale_mode != QATTN_SCALE_HEAD && scale_mode != QATTN_SCALE_TOKEN) return QATTN_ERR_INVALID_ARG;
    if (numerics != QATTN_NUMERICS_COMPILED && numerics != QATTN_NUMERICS_EAGER) return QATTN_ERR_INVALID_ARG;
    if (in_fmt != QATTN_FMT_BF16 && in_fmt != QATTN_FMT_FP16) return QATTN_ERR_UNSUPPORTED_FMT;
    if (fp8_fmt != QATTN_FMT_E4M3 && fp8_fmt != QATTN_FMT_E5M2) return QATTN_ERR_UNSUPPORTED_FMT;
    if (!workspace || workspace_bytes < qattn_quant_qkv_workspace_bytes(B, Hq, Hkv)) return QATTN_ERR_WORKSPACE;
    const bool fuse_q = q_fusion_ok(D, in_fmt, scale_mode);
    unsigned* ws = (unsigned*)workspace;
    int rc = launch_quant_qkv(q, k, v, in_fmt, q8, k8, v8, scale_q, scale_k, scale_v, B, Hq, Hkv, Sq, Skv, D, fp8_fmt, scale_mode,
                              numerics, ws, fuse_q, (hipStream_t)stream);
    if (rc != QATTN_OK) return rc;
    return attention_impl(fuse_q ? nullptr : q8, k8, v8, out, nullptr, fuse_q ? nullptr : scale_q, scale_k, scale_v, B, Hq, Hkv, Sq, Skv,
                          D, fp8_fmt, fp8_fmt, in_fmt, scale_mode, is_causal, sm_scale, fuse_q ? q : nullptr, fuse_q ? ws : nullptr,
                          fuse_q ? scale_q : nullptr, numerics, stream);
}
