// qattn_api.hip -- C-ABI entry points of libqattn_hip.so that are not in qattn_quant.hip (include/qattn.h).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <algorithm>

#include "qattn_attn.h"
#include "qattn_pv16.h"

using namespace qattn;

namespace {

// Per-device events of the bench.py measurement aid (qattn_profile_attention), created on first use.
constexpr int kMaxDevices = 64;
struct DeviceState {
    bool ready = false;
    hipEvent_t prof[2] = {};   // recorded around the attention launches of a call, on their stream
    bool recorded = false;
};
DeviceState g_dev[kMaxDevices];
std::mutex g_mutex;
thread_local int t_profile = 0;

DeviceState* device_state(bool may_create) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    DeviceState& s = g_dev[dev];
    if (s.ready) return &s;
    if (!may_create) return nullptr;
    std::lock_guard<std::mutex> lock(g_mutex);
    if (s.ready) return &s;
    if (hipEventCreate(&s.prof[0]) != hipSuccess || hipEventCreate(&s.prof[1]) != hipSuccess) return nullptr;
    s.ready = true;
    return &s;
}

bool stream_is_capturing(hipStream_t st) {
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(st, &status) == hipSuccess && status != hipStreamCaptureStatusNone;
}

struct AttnCall {
    const void *q8, *k8, *v8;
    void* out;
    float* lse;
    const float *scale_q, *scale_k, *scale_v;
    int B, Hq, Hkv, Sq, Skv, D, qk_fmt, v_fmt, out_fmt, scale_mode, is_causal;
    float sm_scale;
    int precision, lse_layout;
    void* attn_ws;     // nullptr or the attention workspace of THIS call (attn_ws_sched / attn_ws_flags below)
    const unsigned* vexp;         // fused step with a block-scaled V (else nullptr)
    const float *ssq_q, *ssq_k;   // fused step, head-wise AUTO: the heads' partial sums of squares from the pre-pass (else nullptr)
    int ssq_n, ssq_stride;        //   (caller-supplied sums: one entry per head, stride 1)
    const void* q16;   // fused step (else nullptr): bf16 Q, quantised in the kernel from q_amax_part; sq_out is written
    const unsigned* q_amax_part;   // abs-max words of every q head: [B*Hq][amax_stride], amax_n valid per head
    float* sq_out;
    int q_numerics;
    int amax_n, amax_stride;
    const void* v16;              // fused step: the original 16-bit V for the rows that see few keys (else nullptr)
    unsigned long long* stamps;   // measurement entry (else nullptr)
    bool sched_zeroed;            // fused step: the pre-pass has cleared the hand-out counters (else the launch clears them itself)
    unsigned char* path;          // fused entry's per-row path output (else nullptr)
    const long long* q16_strides = nullptr;   // fused step: element strides {batch, head, row} of the 16-bit q / v views (nullptr: dense [B,H,S,D])
    const long long* v16_strides = nullptr;
    const long long* out_strides = nullptr;   // and of `out`
};

// attention workspace = [SchedState of the hand-scheduled kernel's causal launches | one flag word per (b, h, 32-row group)]
size_t attn_ws_sched_bytes(int, int, int) { return (sched_bytes() + 15) / 16 * 16; }
size_t attn_ws_flag_bytes(int B, int Hq, int Sq) { return sizeof(unsigned) * (size_t)B * Hq * ceil_div(Sq, 32); }

// launches the attention kernel(s) of one call on `st`; ds != nullptr: bracket them with the profile events
int attention_impl(const AttnCall& a, hipStream_t st, DeviceState* ds) {
    // argument checks mirror the reference launcher's TORCH_CHECKs (tk/attention.py:362-415)
    if ((!a.q8 && !a.q16) || !a.k8 || !a.v8 || !a.out || (!a.scale_q && !a.q16) || !a.scale_k) return QATTN_ERR_INVALID_ARG;
    if (a.B <= 0 || a.Hq <= 0 || a.Hkv <= 0 || a.Sq <= 0 || a.Skv <= 0) return QATTN_ERR_INVALID_ARG;
    if (a.D != 64 && a.D != 128 && a.D != 256) return QATTN_ERR_UNSUPPORTED_DIM;  // nn.py:45-49
    if (a.Hq % a.Hkv != 0) return QATTN_ERR_UNSUPPORTED_DIM;                      // tk/attention.py:398-399
    if (a.qk_fmt != QATTN_FMT_E4M3 && a.qk_fmt != QATTN_FMT_E5M2) return QATTN_ERR_UNSUPPORTED_FMT;
    // v_fmt: the fp8 format of q / k (both GEMMs on FP8 MFMA, the main path), or a 16-bit format: v8 is then the caller's ROW-MAJOR
    // 16-bit V and every row runs the reference's own P.V numerics (qattn_pv16.h); no two-term / AUTO machinery involved
    const bool v_is_16 = a.v_fmt == QATTN_FMT_BF16 || a.v_fmt == QATTN_FMT_FP16;
    if (!v_is_16 && a.v_fmt != a.qk_fmt) return QATTN_ERR_UNSUPPORTED_FMT;
    if (v_is_16 && (a.q16 != nullptr || a.out_fmt != a.v_fmt)) return QATTN_ERR_UNSUPPORTED_FMT;
    if (a.out_fmt != QATTN_FMT_BF16 && a.out_fmt != QATTN_FMT_FP16) return QATTN_ERR_UNSUPPORTED_FMT;
    if (a.scale_mode != QATTN_SCALE_HEAD && a.scale_mode != QATTN_SCALE_TOKEN) return QATTN_ERR_INVALID_ARG;
    if (a.precision != QATTN_PRECISION_AUTO && a.precision != QATTN_PRECISION_FAST && a.precision != QATTN_PRECISION_ACCURATE) return QATTN_ERR_INVALID_ARG;
    if (a.lse_layout != QATTN_LSE_NATURAL && a.lse_layout != QATTN_LSE_REFERENCE) return QATTN_ERR_INVALID_ARG;
    AttnParams p;
    memset(&p, 0, sizeof(p));
    p.q = (const unsigned char*)a.q8; p.k = (const unsigned char*)a.k8; p.v = (const unsigned char*)a.v8;
    p.out = a.out; p.lse = a.lse; p.sq = a.scale_q; p.sk = a.scale_k; p.sv = a.scale_v;
    p.B = a.B; p.Hq = a.Hq; p.Hkv = a.Hkv; p.Sq = a.Sq; p.Skv = a.Skv;
    p.nqb = ceil_div(a.Sq, kQPerWG);
    p.nchunks = ceil_div(a.Skv, 64);
    p.out_fmt = a.out_fmt;
    p.xcd_remap = ((a.B * a.Hq) % 8 == 0 && xcd_count() == 8) ? 1 : 0;   // (the maps of qattn_attn.h are written for 8 XCDs)
    // causal: heads taken in groups whose K + V (2 Skv D bytes of fp8 per head) stay within half of an XCD's 4 MiB L2 -- with
    // 4 heads at S = 4096 (4 MiB) the PMC passes counted 1.3x the algorithmic HBM bytes, at S = 16384 (16 MiB) 2.4x
    p.causal_group = 1;
    for (int g = kCausalHeadGroup; g > 1; g >>= 1)
        if (p.xcd_remap && ((a.B * a.Hq) >> 3) % g == 0 && (size_t)g * 2 * a.Skv * a.D <= kCausalGroupBytes) { p.causal_group = g; break; }
    // causal AUTO: the blocks right above the two-term line first (causal_order, qattn_attn.h)
    // causal, XCD-aware hand-out: every head's blocks below the two-term line wait for the end of the XCD's list (map_block)
    {
        const int early = (kTwoTermKeys + kQPerWG - 2) / kQPerWG;
#ifndef QATTN_TAIL_LO
#define QATTN_TAIL_LO 1   // (a build knob for tools/ab.py variants: 0 = every group's own blocks, round 4's order)
#endif
        p.tail_lo = (QATTN_TAIL_LO != 0 && a.is_causal && p.xcd_remap && p.nqb >= 2 * early) ? early : 0;
    }
    p.risky_lo = p.risky_hi = 0;
    if (a.is_causal && a.precision == QATTN_PRECISION_AUTO) {
        p.risky_lo = std::min(p.nqb, (kTwoTermKeys + kQPerWG - 2) / kQPerWG);   // blocks whose first row sees < kTwoTermKeys keys: qb < lo
        p.risky_hi = std::min(p.nqb, 2 * p.risky_lo);
    }
    const float sm = a.sm_scale > 0.0f ? a.sm_scale : 1.0f / sqrtf((float)a.D);
    p.sm_log2e = sm * 1.4426950408889634f;
    p.precision = a.precision;
    p.two_term_keys = kTwoTermKeys;
    p.peak_r0 = a.precision == QATTN_PRECISION_AUTO ? kPeakR0 : 0.0f;
    p.peak_neff = a.precision == QATTN_PRECISION_AUTO ? kPeakNeff : 0.0f;
    p.max_rescue = kMaxRescueWaves;
    p.max_rescue_rows = kMaxRescueRows;
    p.persistent = 1;
    p.dyn_min_rounds = kDynMinRounds;
    p.ssq_q = a.precision == QATTN_PRECISION_AUTO ? a.ssq_q : nullptr;   // (FAST: the caller vouches for flat rows)
    p.ssq_k = p.ssq_q ? a.ssq_k : nullptr;
    p.vexp = a.vexp;
    p.ssq_n = a.ssq_n; p.ssq_stride = a.ssq_stride;
    p.amax_n = a.amax_n; p.amax_stride = a.amax_stride; p.vexp_stride = kMomentSplits;
    p.var_mul = sm * sm / ((float)a.Sq * (float)a.Skv * (float)a.D);
    p.sched = (SchedState*)a.attn_ws;
    p.sched_zeroed = a.sched_zeroed ? 1 : 0;
    p.flags = a.attn_ws ? (unsigned*)((unsigned char*)a.attn_ws + attn_ws_sched_bytes(a.B, a.Hq, a.Sq)) : nullptr;
    p.lse_stride = (long)qattn_lse_row_stride(a.Sq, a.lse_layout);
    p.lse_mul = a.lse_layout == QATTN_LSE_REFERENCE ? -sqrtf((float)a.D) : 1.0f;
    p.q16 = (const unsigned char*)a.q16; p.q_amax_part = a.q_amax_part; p.sq_out = a.sq_out; p.q_numerics = a.q_numerics;
    p.stamp_buf = a.stamps;
    p.path = a.path;
    p.v16 = v_is_16 ? (const unsigned char*)a.v8 : (const unsigned char*)a.v16;
    // byte strides of the 16-bit Q / V as the kernels read them (qattn_attn.h q16_row / v16_head): the caller's view, or dense
    p.q16_rs = 2L * a.D; p.q16_hs = p.q16_rs * a.Sq; p.q16_bs = p.q16_hs * a.Hq;
    p.v16_rs = 2L * a.D; p.v16_hs = p.v16_rs * a.Skv; p.v16_bs = p.v16_hs * a.Hkv;
    if (a.q16 && a.q16_strides) { p.q16_bs = 2 * a.q16_strides[0]; p.q16_hs = 2 * a.q16_strides[1]; p.q16_rs = 2 * a.q16_strides[2]; }
    if (!v_is_16 && a.v16 && a.v16_strides) { p.v16_bs = 2 * a.v16_strides[0]; p.v16_hs = 2 * a.v16_strides[1]; p.v16_rs = 2 * a.v16_strides[2]; }
    p.o_rs = 2L * a.D; p.o_hs = p.o_rs * a.Sq; p.o_bs = p.o_hs * a.Hq;
    if (a.out_strides) { p.o_bs = 2 * a.out_strides[0]; p.o_hs = 2 * a.out_strides[1]; p.o_rs = 2 * a.out_strides[2]; }
    if (a.stamps && !(a.q16 && attn_v2_covers(a.D, a.is_causal, a.scale_mode) && a.qk_fmt == QATTN_FMT_E4M3)) return QATTN_ERR_UNSUPPORTED_FMT;
    const bool use_v2 = attn_v2_covers(a.D, a.is_causal, a.scale_mode);
    p.peak_z = (float)p.two_term_keys > kPeakR0 ? 0.5f + logf((float)p.two_term_keys / kPeakR0) : 0.0f;   // see predicted_r
    const bool prof = ds != nullptr;
    if (prof) (void)hipEventRecord(ds->prof[0], st);
    // (the fused step on the D = 128 kernel always quantises Q in the kernel -- q_fusion_ok -- and carries its 16-bit-V pass inside; the
    // templated kernel's fused calls fork the launch of their early rows onto the side stream in launch_v4_full_d)
    int rc;
    if (v_is_16) rc = launch_attn_pv16(p, a.D, a.qk_fmt, a.v_fmt, a.is_causal, a.scale_mode, st);
    else if (use_v2) rc = launch_attn_v2(p, a.D, a.qk_fmt, a.is_causal, a.scale_mode, st);
    else rc = launch_attn_v4_full(p, a.D, a.qk_fmt, a.is_causal, a.scale_mode, st);
    if (prof) { (void)hipEventRecord(ds->prof[1], st); ds->recorded = true; }
    if (rc != QATTN_OK) return rc;
    return hipGetLastError() == hipSuccess ? QATTN_OK : QATTN_ERR_LAUNCH;
}

// true when the attention kernel can quantise Q itself (hand-scheduled D = 128 kernel, head-wise scales, bf16 or -- since round 5 -- fp16
// inputs, byte-exponential path): then the pre-pass skips Q's payload (one read and one write of Q less).
bool q_fusion_ok(int D, int in_fmt, int scale_mode, int is_causal) {
    return D == 128 && (in_fmt == QATTN_FMT_BF16 || in_fmt == QATTN_FMT_FP16) && scale_mode == QATTN_SCALE_HEAD && attn_v2_covers(D, is_causal, scale_mode);
}

// fused entry: block-scaled V (one power-of-two scale per 64-key chunk) wherever the kernel's PV products take the chunk's scale byte and a
// head has at most kMomentSplits chunks (quant_attention_impl; qattn_describe_path reports the same predicate)
bool fused_v_block(bool fuse_q, int D, int scale_mode, int is_causal, int Skv) {
    const bool vs_kernel = fuse_q || (scale_mode == QATTN_SCALE_HEAD && !attn_v2_covers(D, is_causal, scale_mode));
    return vs_kernel && (Skv + 63) / 64 <= kMomentSplits;
}

}  // namespace

namespace qattn {
// ---- side stream (declared in qattn_attn.h)
struct SideStream {
    bool ready = false, failed = false;
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    void release() {
        if (fork) (void)hipEventDestroy(fork);
        if (join) (void)hipEventDestroy(join);
        if (s) (void)hipStreamDestroy(s);
        fork = join = nullptr;
        s = nullptr;
        ready = false;
    }
    // a host thread that exits gives its stream and events back (thread-pool servers).  (At process exit the runtime may be
    // gone already: the destroy calls then fail harmlessly.)
    ~SideStream() { release(); }
};
thread_local SideStream t_side[kMaxDevices];

#ifndef QATTN_SIDE_STREAM
#define QATTN_SIDE_STREAM 1   // (a build knob for tools/ab.py variants: 0 = every launch of a call on the caller's stream)
#endif
hipStream_t side_stream_fork(hipStream_t st) {
    if (!QATTN_SIDE_STREAM) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    SideStream& x = t_side[dev];
    if (x.failed) return nullptr;   // creation failed once on this thread: single-stream from then on, no retry (and no leak per call)
    if (!x.ready) {
        if (stream_is_capturing(st)) return nullptr;   // (no object creation inside a capture)
        if (hipStreamCreateWithFlags(&x.s, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&x.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&x.join, hipEventDisableTiming) != hipSuccess) {
            x.release();   // whatever was created
            x.failed = true;
            (void)hipGetLastError();
            return nullptr;
        }
        x.ready = true;
    }
    if (hipEventRecord(x.fork, st) != hipSuccess || hipStreamWaitEvent(x.s, x.fork, 0) != hipSuccess) return nullptr;
    return x.s;
}
int side_stream_join(hipStream_t st, hipStream_t side) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return QATTN_ERR_LAUNCH;
    SideStream& x = t_side[dev];
    if (!x.ready || side != x.s) return QATTN_ERR_LAUNCH;
    if (hipEventRecord(x.join, side) != hipSuccess || hipStreamWaitEvent(st, x.join, 0) != hipSuccess) return QATTN_ERR_LAUNCH;
    return QATTN_OK;
}

__global__ void fill_bytes_kernel(unsigned char* w, long n, unsigned char v) {   // the row_path output's pre-fill (a kernel node: see zero_words)
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] = v;
}
__global__ void zero_words_kernel(unsigned* w, long n) {   // (declared in qattn_attn.h: zero_words)
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] = 0u;
}
}  // namespace qattn

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

extern "C" void qattn_profile_attention(int enable) { t_profile = enable != 0; }

extern "C" float qattn_last_attention_ms(void) {
    DeviceState* ds = device_state(false);
    if (!ds || !ds->recorded) return -1.0f;
    float ms = 0.0f;
    if (hipEventSynchronize(ds->prof[1]) != hipSuccess || hipEventElapsedTime(&ms, ds->prof[0], ds->prof[1]) != hipSuccess) return -1.0f;
    return ms;
}

extern "C" size_t qattn_attention_workspace_bytes(int B, int Hq, int Sq) {
    if (B <= 0 || Hq <= 0 || Sq <= 0) return 0;
    return attn_ws_sched_bytes(B, Hq, Sq) + attn_ws_flag_bytes(B, Hq, Sq);
}

extern "C" size_t qattn_lse_row_stride(int Sq, int lse_layout) {
    if (Sq <= 0) return 0;
    return lse_layout == QATTN_LSE_REFERENCE ? ((size_t)Sq * 4 + 15) / 16 * 16 / 4 : (size_t)Sq;  // tk/attention.py:439
}

extern "C" int qattn_fp8_attention_forward(const void* q8, const void* k8, const void* v8, void* out, float* lse,
                                           const float* scale_q, const float* scale_k, const float* scale_v, int B,
                                           int Hq, int Hkv, int Sq, int Skv, int D, int qk_fmt, int v_fmt, int out_fmt,
                                           int scale_mode, int is_causal, float sm_scale, int precision, int lse_layout,
                                           void* workspace, size_t workspace_bytes, void* stream) {
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || Sq <= 0 || Skv <= 0) return QATTN_ERR_INVALID_ARG;
    if ((D != 64 && D != 128 && D != 256) || Hq % Hkv != 0) return QATTN_ERR_UNSUPPORTED_DIM;   // (before the size of the workspace is judged)
    const bool have_ws = workspace && workspace_bytes >= qattn_attention_workspace_bytes(B, Hq, Sq);
    if (precision == QATTN_PRECISION_AUTO && !have_ws) return QATTN_ERR_WORKSPACE;
    AttnCall a{q8, k8, v8, out, lse, scale_q, scale_k, scale_v, B, Hq, Hkv, Sq, Skv, D, qk_fmt, v_fmt, out_fmt, scale_mode,
               is_causal, sm_scale, precision, lse_layout, have_ws ? workspace : nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, nullptr, false, nullptr};
    DeviceState* ds = t_profile ? device_state(!stream_is_capturing((hipStream_t)stream)) : nullptr;
    return attention_impl(a, (hipStream_t)stream, ds);
}

extern "C" size_t qattn_fp8_quant_attention_workspace_bytes(int B, int Hq, int Hkv, int Sq) {
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || Sq <= 0) return 0;
    // [abs-max bits of q, k, v | the attention call's workspace], the second part 16-byte aligned
    return (qattn_quant_qkv_workspace_bytes(B, Hq, Hkv) + 15) / 16 * 16 + qattn_attention_workspace_bytes(B, Hq, Sq);
}

static int quant_attention_impl(const void* q, const void* k, const void* v, int in_fmt, void* out, void* q8,
                                void* k8, void* v8, float* scale_q, float* scale_k, float* scale_v,
                                const float* amax_q, const float* amax_k, const float* amax_v,
                                const float* ssq_q, const float* ssq_k, int B, int Hq, int Hkv, int Sq, int Skv,
                                int D, int fp8_fmt, int scale_mode, int numerics, int is_causal, float sm_scale,
                                int precision, float* lse, int lse_layout, unsigned char* row_path, void* workspace, size_t workspace_bytes,
                                void* stream, unsigned long long* stamps, const long long* strides = nullptr) {
    if (!q || !k || !v || !out || !q8 || !k8 || !v8 || !scale_q || !scale_k || !scale_v) return QATTN_ERR_INVALID_ARG;
    if (lse_layout != QATTN_LSE_NATURAL && lse_layout != QATTN_LSE_REFERENCE) return QATTN_ERR_INVALID_ARG;
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || Sq <= 0 || Skv <= 0) return QATTN_ERR_INVALID_ARG;
    if (D != 64 && D != 128 && D != 256) return QATTN_ERR_UNSUPPORTED_DIM;
    if (Hq % Hkv != 0) return QATTN_ERR_UNSUPPORTED_DIM;
    if (scale_mode != QATTN_SCALE_HEAD && scale_mode != QATTN_SCALE_TOKEN) return QATTN_ERR_INVALID_ARG;
    if (numerics != QATTN_NUMERICS_COMPILED && numerics != QATTN_NUMERICS_EAGER) return QATTN_ERR_INVALID_ARG;
    if (in_fmt != QATTN_FMT_BF16 && in_fmt != QATTN_FMT_FP16) return QATTN_ERR_UNSUPPORTED_FMT;
    if (fp8_fmt != QATTN_FMT_E4M3 && fp8_fmt != QATTN_FMT_E5M2) return QATTN_ERR_UNSUPPORTED_FMT;
    if ((amax_q || amax_k || amax_v || ssq_q || ssq_k) && scale_mode != QATTN_SCALE_HEAD) return QATTN_ERR_INVALID_ARG;   // per-head figures
    if ((ssq_q == nullptr) != (ssq_k == nullptr)) return QATTN_ERR_INVALID_ARG;
    if (strides) {
        // strided views of the 16-bit inputs (qattn_fp8_quant_attention_forward_strided): D innermost and dense, every row 16-byte aligned,
        // no two rows overlapping is the caller's business; 64 rows of a tensor within 2^31 bytes (32-bit lane offsets of the LDS-DMA requests)
        const void* base[4] = {q, k, v, out};
        for (int t = 0; t < 4; t++) {
            if ((reinterpret_cast<uintptr_t>(base[t]) & 15u) != 0) return QATTN_ERR_INVALID_ARG;
            for (int i = 0; i < 3; i++)
                if (strides[3 * t + i] < 0 || strides[3 * t + i] % 8 != 0) return QATTN_ERR_INVALID_ARG;
            if (strides[3 * t + 2] < D || strides[3 * t + 2] > (1LL << 23)) return QATTN_ERR_INVALID_ARG;
        }
        if ((B > 1 && strides[9] == 0) || (Hq > 1 && strides[10] == 0)) return QATTN_ERR_INVALID_ARG;   // (`out` cannot be a broadcast view)
    }
    if (!workspace || workspace_bytes < qattn_fp8_quant_attention_workspace_bytes(B, Hq, Hkv, Sq)) return QATTN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const bool fuse_q = q_fusion_ok(D, in_fmt, scale_mode, is_causal);
    // the measurement entry exists for one instantiation only: refuse before the pre-pass has written anything
    // (the stamped instantiation exists for bf16 Q rows only: ADVICE r5)
    if (stamps && !(fuse_q && fp8_fmt == QATTN_FMT_E4M3 && in_fmt == QATTN_FMT_BF16)) return QATTN_ERR_UNSUPPORTED_FMT;
    // (Tried and dropped, profiles/r02_overlap.md: running the HBM-bound pre-pass of batch group g+1 on a second stream beside
    // the attention of group g.  The 512-thread attention workgroups leave 32-48 VGPRs per SIMD, the pre-pass waves displace
    // them instead of sharing the CU, and the chip is power-limited on the attention kernel: the step got 19-31 % SLOWER.)
    unsigned* ws = (unsigned*)workspace;
    void* attn_ws = (unsigned char*)workspace + (qattn_quant_qkv_workspace_bytes(B, Hq, Hkv) + 15) / 16 * 16;
    // the score-spread estimate needs both heads' sums of squares: from the abs-max pass when it reads q AND k, else from the caller
    const bool auto_head = precision == QATTN_PRECISION_AUTO && scale_mode == QATTN_SCALE_HEAD;
    const bool ext_moments = auto_head && ssq_q != nullptr;
    // (a caller that hands over only ONE of amax_q / amax_k without the sums keeps the estimate: both tensors then still go
    // through the abs-max pass for their sums of squares -- the supplied abs-max is used for the scale, nothing is saved, and the
    // call produces the plain call's bits; with BOTH supplied and no sums the pass is skipped and wide heads start one-term)
    const bool moments = auto_head && !ext_moments && !(amax_q && amax_k);
    // block-scaled V with head-wise scales wherever the kernel's PV products take the chunk's scale byte -- the hand-scheduled
    // D = 128 kernel in its fused-Q instantiation, the templated kernel (D = 64 / 256) -- and a head has at most kMomentSplits
    // chunks: V then needs no abs-max pass
    const bool v_block = fused_v_block(fuse_q, D, scale_mode, is_causal, Skv);
    const float* ext_amax[3] = {amax_q, amax_k, amax_v};
    // the hand-out counters of the attention launch (D = 128 kernel) are cleared by the quantise pass on its way: a launch of
    // its own for 32 bytes sat between the two kernels for ~5 us
    // (the templated kernel's AUTO launches start from cleared peaked-group flags, which live behind the counters: cleared on the same way)
    const bool zero_in_prepass = attn_ws != nullptr;
    const size_t zero_bytes = attn_v2_covers(D, is_causal, scale_mode) ? sched_bytes()
                              : precision == QATTN_PRECISION_AUTO ? attn_ws_sched_bytes(B, Hq, Sq) + attn_ws_flag_bytes(B, Hq, Sq) : 0;
    int rc = launch_quant_qkv(q, k, v, in_fmt, q8, k8, v8, scale_q, scale_k, scale_v, B, Hq, Hkv, Sq, Skv, D, fp8_fmt, scale_mode,
                              numerics, ws, fuse_q, moments, v_block, st, ext_amax, zero_in_prepass && zero_bytes ? (unsigned*)attn_ws : nullptr,
                              zero_in_prepass ? (int)(zero_bytes / sizeof(unsigned)) : 0, strides);
    if (rc != QATTN_OK) return rc;
    if (row_path) {   // every row starts as "one-term fp8-V sweep"; the other passes overwrite what they store
        const long n = (long)B * Hq * Sq;
        hipLaunchKernelGGL(fill_bytes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, row_path, n, (unsigned char)QATTN_PATH_ONE_TERM);
        if (hipGetLastError() != hipSuccess) return QATTN_ERR_LAUNCH;
    }
    const QuantMoments mom = quant_moments(ws, B, Hq, Hkv, Sq, Skv, D);
    const bool q_ext = amax_q != nullptr;
    AttnCall a{fuse_q ? nullptr : q8, k8, v8, out, lse, fuse_q ? nullptr : scale_q, scale_k, scale_v, B, Hq, Hkv, Sq, Skv, D,
               fp8_fmt, fp8_fmt, in_fmt, scale_mode, is_causal, sm_scale, precision, lse_layout, attn_ws,
               v_block ? mom.vexp : nullptr,
               moments ? mom.part_q : ext_moments ? ssq_q : nullptr, moments ? mom.part_k : ext_moments ? ssq_k : nullptr,
               ext_moments ? 1 : mom.nsplit, ext_moments ? 1 : kMomentSplits,
               fuse_q ? q : nullptr, fuse_q ? (q_ext ? reinterpret_cast<const unsigned*>(amax_q) : mom.amax_q) : nullptr, fuse_q ? scale_q : nullptr, numerics,
               q_ext ? 1 : mom.nsplit, q_ext ? 1 : kMomentSplits, v, stamps, zero_in_prepass, row_path, strides, strides ? strides + 6 : nullptr, strides ? strides + 9 : nullptr};
    DeviceState* ds = t_profile ? device_state(!stream_is_capturing(st)) : nullptr;
    return attention_impl(a, st, ds);
}

extern "C" int qattn_fp8_quant_attention_forward_ex(const void* q, const void* k, const void* v, int in_fmt, void* out, void* q8,
                                                    void* k8, void* v8, float* scale_q, float* scale_k, float* scale_v,
                                                    const float* amax_q, const float* amax_k, const float* amax_v,
                                                    const float* ssq_q, const float* ssq_k, int B, int Hq, int Hkv, int Sq, int Skv,
                                                    int D, int fp8_fmt, int scale_mode, int numerics, int is_causal, float sm_scale,
                                                    int precision, float* lse, int lse_layout, unsigned char* row_path, void* workspace,
                                                    size_t workspace_bytes, void* stream) {
    return quant_attention_impl(q, k, v, in_fmt, out, q8, k8, v8, scale_q, scale_k, scale_v, amax_q, amax_k, amax_v, ssq_q, ssq_k, B, Hq, Hkv,
                                Sq, Skv, D, fp8_fmt, scale_mode, numerics, is_causal, sm_scale, precision, lse, lse_layout, row_path, workspace,
                                workspace_bytes, stream, nullptr);
}

// The fused entry on STRIDED VIEWS of the 16-bit inputs -- e.g. q = x.view(B, S, H, D).transpose(1, 2), the form attention inputs have in
// most callers.  The reference reads such q / k in its Inductor-made quantiser and copies such a v (`.contiguous()`, tk/attention.py:419-421);
// here every kernel that touches the 16-bit tensors (abs-max pass, quantise pass, the attention kernels' Q rows and 16-bit-V passes) takes
// the strides, so nothing is copied and the result is the dense call's, bit for bit.
extern "C" int qattn_fp8_quant_attention_forward_strided(const void* q, const void* k, const void* v, const long long* strides, int in_fmt, void* out,
                                                         void* q8, void* k8, void* v8, float* scale_q, float* scale_k, float* scale_v,
                                                         const float* amax_q, const float* amax_k, const float* amax_v,
                                                         const float* ssq_q, const float* ssq_k, int B, int Hq, int Hkv, int Sq, int Skv,
                                                         int D, int fp8_fmt, int scale_mode, int numerics, int is_causal, float sm_scale,
                                                         int precision, float* lse, int lse_layout, unsigned char* row_path, void* workspace,
                                                         size_t workspace_bytes, void* stream) {
    return quant_attention_impl(q, k, v, in_fmt, out, q8, k8, v8, scale_q, scale_k, scale_v, amax_q, amax_k, amax_v, ssq_q, ssq_k, B, Hq, Hkv,
                                Sq, Skv, D, fp8_fmt, scale_mode, numerics, is_causal, sm_scale, precision, lse, lse_layout, row_path, workspace,
                                workspace_bytes, stream, nullptr, strides);
}

extern "C" size_t qattn_attention_stamp_bytes(int B, int Hq, int Sq) {
    if (B <= 0 || Hq <= 0 || Sq <= 0) return 0;
    return 2 * sizeof(unsigned long long) * (size_t)B * Hq * ceil_div(Sq, kQPerWG) * kWaves;
}

extern "C" int qattn_fp8_quant_attention_forward_stamped(const void* q, const void* k, const void* v, int in_fmt, void* out, void* q8,
                                                         void* k8, void* v8, float* scale_q, float* scale_k, float* scale_v, int B,
                                                         int Hq, int Hkv, int Sq, int Skv, int D, int fp8_fmt, int scale_mode,
                                                         int numerics, int is_causal, float sm_scale, int precision, void* workspace,
                                                         size_t workspace_bytes, void* stamps, size_t stamps_bytes, void* stream) {
    if (!stamps || stamps_bytes < qattn_attention_stamp_bytes(B, Hq, Sq)) return QATTN_ERR_WORKSPACE;
    return quant_attention_impl(q, k, v, in_fmt, out, q8, k8, v8, scale_q, scale_k, scale_v, nullptr, nullptr, nullptr, nullptr, nullptr, B, Hq,
                                Hkv, Sq, Skv, D, fp8_fmt, scale_mode, numerics, is_causal, sm_scale, precision, nullptr, QATTN_LSE_NATURAL, nullptr,
                                workspace, workspace_bytes, stream, (unsigned long long*)stamps);
}

extern "C" int qattn_fp8_quant_attention_forward(const void* q, const void* k, const void* v, int in_fmt, void* out, void* q8,
                                                 void* k8, void* v8, float* scale_q, float* scale_k, float* scale_v, int B,
                                                 int Hq, int Hkv, int Sq, int Skv, int D, int fp8_fmt, int scale_mode,
                                                 int numerics, int is_causal, float sm_scale, int precision, void* workspace,
                                                 size_t workspace_bytes, void* stream) {
    return qattn_fp8_quant_attention_forward_ex(q, k, v, in_fmt, out, q8, k8, v8, scale_q, scale_k, scale_v, nullptr, nullptr, nullptr,
                                                nullptr, nullptr, B, Hq, Hkv, Sq, Skv, D, fp8_fmt, scale_mode, numerics, is_causal,
                                                sm_scale, precision, nullptr, QATTN_LSE_NATURAL, nullptr, workspace, workspace_bytes, stream);
}

// ---- the pybind function's contract in ONE call (tk/attention.py:357-360, 419-437): row-major fp8 q / k, 16-bit v, fp32 scales -> out
extern "C" size_t qattn_fp8_attention_rowmajor_workspace_bytes(int B, int Hq, int Hkv, int Sq, int Skv, int D) {
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || Sq <= 0 || Skv <= 0 || (D != 64 && D != 128 && D != 256)) return 0;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    return up(qattn_fp8_tensor_bytes(QATTN_LAYOUT_KFRAG, B, Hkv, Skv, D)) + up(qattn_fp8_tensor_bytes(QATTN_LAYOUT_VFRAG, B, Hkv, Skv, D)) +
           up(sizeof(float) * (size_t)B * Hkv) + up(qattn_quant_workspace_bytes(B, Hkv, Skv, D, QATTN_SCALE_HEAD)) + up(qattn_attention_workspace_bytes(B, Hq, Sq));
}

extern "C" int qattn_fp8_attention_forward_rowmajor(const void* q8, const void* k8, const void* v16, void* out, float* lse, const float* scale_q,
                                                    const float* scale_k, int B, int Hq, int Hkv, int Sq, int Skv, int D, int qk_fmt, int v16_fmt,
                                                    int pv_fmt, int scale_mode, int is_causal, float sm_scale, int precision, int lse_layout,
                                                    void* workspace, size_t workspace_bytes, void* stream) {
    if (!q8 || !k8 || !v16 || !out || !scale_q || !scale_k) return QATTN_ERR_INVALID_ARG;
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || Sq <= 0 || Skv <= 0) return QATTN_ERR_INVALID_ARG;
    if ((D != 64 && D != 128 && D != 256) || Hq % Hkv != 0) return QATTN_ERR_UNSUPPORTED_DIM;
    if (qk_fmt != QATTN_FMT_E4M3 && qk_fmt != QATTN_FMT_E5M2) return QATTN_ERR_UNSUPPORTED_FMT;
    if (v16_fmt != QATTN_FMT_BF16 && v16_fmt != QATTN_FMT_FP16) return QATTN_ERR_UNSUPPORTED_FMT;
    if (pv_fmt != qk_fmt && pv_fmt != v16_fmt) return QATTN_ERR_UNSUPPORTED_FMT;   // both GEMMs on FP8 MFMA, or the reference's 16-bit P.V
    if (!workspace || workspace_bytes < qattn_fp8_attention_rowmajor_workspace_bytes(B, Hq, Hkv, Sq, Skv, D)) return QATTN_ERR_WORKSPACE;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    unsigned char* w = (unsigned char*)workspace;
    void* kfrag = w;                w += up(qattn_fp8_tensor_bytes(QATTN_LAYOUT_KFRAG, B, Hkv, Skv, D));
    void* vfrag = w;                w += up(qattn_fp8_tensor_bytes(QATTN_LAYOUT_VFRAG, B, Hkv, Skv, D));
    float* scale_v = (float*)w;     w += up(sizeof(float) * (size_t)B * Hkv);
    void* quant_ws = w;             const size_t quant_ws_bytes = qattn_quant_workspace_bytes(B, Hkv, Skv, D, QATTN_SCALE_HEAD);
    w += up(quant_ws_bytes);
    void* attn_ws = w;              const size_t attn_ws_bytes = qattn_attention_workspace_bytes(B, Hq, Sq);
    // K: a byte permutation into the MFMA fragment order (the reference launcher's `.contiguous()`, tk/attention.py:419-421, is the analogue)
    int rc = qattn_pack_fp8(k8, kfrag, B, Hkv, Skv, D, QATTN_LAYOUT_KFRAG, stream);
    if (rc != QATTN_OK) return rc;
    if (pv_fmt == qk_fmt) {   // V quantised head-wise with the reference quantiser's sequence (nn.py:14-19) straight into fragment order
        rc = qattn_quant_fp8(v16, v16_fmt, vfrag, scale_v, B, Hkv, Skv, D, qk_fmt, QATTN_SCALE_HEAD, QATTN_NUMERICS_COMPILED, QATTN_LAYOUT_VFRAG,
                             quant_ws, quant_ws_bytes, stream);
        if (rc != QATTN_OK) return rc;
        return qattn_fp8_attention_forward(q8, kfrag, vfrag, out, lse, scale_q, scale_k, scale_v, B, Hq, Hkv, Sq, Skv, D, qk_fmt, qk_fmt, v16_fmt,
                                           scale_mode, is_causal, sm_scale, precision, lse_layout, attn_ws, attn_ws_bytes, stream);
    }
    return qattn_fp8_attention_forward(q8, kfrag, v16, out, lse, scale_q, scale_k, nullptr, B, Hq, Hkv, Sq, Skv, D, qk_fmt, v16_fmt, v16_fmt, scale_mode,
                                       is_causal, sm_scale, precision, lse_layout, attn_ws, attn_ws_bytes, stream);
}

// ---- which numerics an entry runs for given arguments: host-only, from the very predicates the dispatch above uses
extern "C" int qattn_describe_path(int entry, int D, int in_fmt, int scale_mode, int Skv, int want_lse, qattn_path_desc* d) {
    if (!d) return QATTN_ERR_INVALID_ARG;
    if (entry != QATTN_ENTRY_SEPARATE && entry != QATTN_ENTRY_SEPARATE_V16 && entry != QATTN_ENTRY_FUSED) return QATTN_ERR_INVALID_ARG;
    if (D != 64 && D != 128 && D != 256) return QATTN_ERR_UNSUPPORTED_DIM;
    if (scale_mode != QATTN_SCALE_HEAD && scale_mode != QATTN_SCALE_TOKEN) return QATTN_ERR_INVALID_ARG;
    if (in_fmt != QATTN_FMT_BF16 && in_fmt != QATTN_FMT_FP16) return QATTN_ERR_UNSUPPORTED_FMT;
    if (Skv <= 0) return QATTN_ERR_INVALID_ARG;
    memset(d, 0, sizeof(*d));
    const bool v2 = attn_v2_covers(D, 0, scale_mode);
    if (entry == QATTN_ENTRY_SEPARATE_V16) {   // attention_impl: v_is_16 -> launch_attn_pv16 for every row
        d->kernel = QATTN_KERNEL_PV16; d->q_quant = QATTN_QQUANT_CALLER; d->v_format = QATTN_VFORMAT_16BIT; d->sweep_p = QATTN_SWEEP_P16;
        d->precise = QATTN_PRECISE_NONE; d->early = QATTN_EARLY_NONE; d->start_mode = QATTN_START_NONE; d->lse = QATTN_LSE_SRC_EXACT;
        return QATTN_OK;
    }
    d->kernel = v2 ? QATTN_KERNEL_V2 : QATTN_KERNEL_V4;
    if (entry == QATTN_ENTRY_SEPARATE) {   // qattn_fp8_attention_forward / _rowmajor with an fp8 V: no 16-bit V at hand
        d->q_quant = QATTN_QQUANT_CALLER; d->v_format = QATTN_VFORMAT_HEAD;
        d->sweep_p = want_lse ? QATTN_SWEEP_EXACT : QATTN_SWEEP_BYTE;   // launch_attn_v2_t / launch_v4_full_d: byte_exp = lse == nullptr
        d->precise = QATTN_PRECISE_TWO_TERM; d->early = QATTN_EARLY_TWO_TERM; d->start_mode = QATTN_START_KEYS; d->lse = QATTN_LSE_SRC_EXACT;
        return QATTN_OK;
    }
    const bool fuse_q = q_fusion_ok(D, in_fmt, scale_mode, 0);
    d->q_quant = fuse_q ? QATTN_QQUANT_KERNEL : QATTN_QQUANT_PREPASS;
    d->v_format = fused_v_block(fuse_q, D, scale_mode, 0, Skv) ? QATTN_VFORMAT_BLOCK : QATTN_VFORMAT_HEAD;
    // the fused D = 128 kernel is byte-exponential whatever is asked (launch_attn_v2_t: p.q16 != nullptr); the templated one switches
    d->sweep_p = (fuse_q || !want_lse) ? QATTN_SWEEP_BYTE : QATTN_SWEEP_EXACT;
    d->precise = fuse_q ? QATTN_PRECISE_V16 : QATTN_PRECISE_TWO_TERM;            // run_block kPass16 / launch_v4_full_d's two-term launches
    d->early = fuse_q ? QATTN_EARLY_V16_INLINE : QATTN_EARLY_V16_LAUNCH;          // pv16p_block_pass inside the kernel / launch_attn_pv16 beside it
    d->start_mode = scale_mode == QATTN_SCALE_HEAD ? QATTN_START_MOMENTS : QATTN_START_KEYS;   // quant_attention_impl: auto_head
    d->lse = fuse_q ? QATTN_LSE_SRC_QUANTISED : QATTN_LSE_SRC_EXACT;
    return QATTN_OK;
}
