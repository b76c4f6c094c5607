// qattn_attn_v1.hip -- FP8 fused attention forward, first (non-pipelined) structure.  Kept as the D=256 path and
// as the A/B baseline for qattn_attn_v2.hip (select with QATTN_KERNEL_VARIANT=1).
//
// Replaces fwd_attend_ker<D,causal,..> + its launcher (src/quantum_attn/tk/attention.py:97-349, 355-647) behind the
// op quantum_attn::fp8_attention_forward (src/quantum_attn/ops.py:98-121).  Designed for CDNA4, not translated:
//
//  * workgroup = 8 waves (2 per SIMD) = 256 query rows; each wave owns 32 query rows for the whole KV sweep.
//  * both GEMMs on v_mfma_f32_32x32x64_f8f6f4 (unscaled form = full FP8 rate, profiles/r01_mfma_probe.log):
//        S^T[key][q] = K . Q^T      (A = K fragment from LDS, B = Q^T fragment held in registers)
//        O^T[d][q]  += V^T . P^T    (A = V^T fragment from LDS, B = P^T built in registers from S^T)
//    The swapped orientation puts the query on the LANE (col = lane&31) and the keys in the accumulator registers
//    (row = (r&3) + 8*(r>>2) + 4*(lane>>5)), so the softmax statistics are per-lane scalars, the only cross-lane
//    traffic is one v_permlane32_swap per chunk, and the fp8-converted P registers ARE the next MFMA's B operand.
//  * K and V arrive pre-laid in fragment order (include/qattn.h) so a 64-key chunk is one linear 2*64*D-byte
//    LDS-DMA copy (global_load_lds_dwordx4) and every operand read is a conflict-free ds_read_b128.
//  * 3-stage LDS ring, one s_barrier per chunk, DMA two chunks ahead behind a counted vmcnt.
//  * online softmax in the exp2 domain with a deferred-max rescale (O and l are only rescaled when some row's
//    max grew by more than kRescaleThr); P is scaled by 2^kPShift before the e4m3 conversion to keep small
//    probabilities out of the subnormal range; where few keys are visible P is split hi+lo (two fp8 terms).
#include "qattn_attn.h"

namespace qattn {

constexpr int kStages = 3;





template <int D>
__device__ inline void stage_chunk(const unsigned char* kg, const unsigned char* vg, unsigned char* lds_stage, int wave, int lane) {
    // [K chunk | V chunk] = 2*64*D bytes, linear; every wave-instruction moves 1 KiB (64 lanes x 16 B)
    constexpr int CH = 64 * D;
    constexpr int ROUNDS = 2 * CH / (kThreads * 16);
    const int wave_base = wave << 10;  // wave-uniform (wave came through readfirstlane)
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const int o = r * (kThreads * 16) + wave_base;  // wave-uniform byte offset in the stage image
        const unsigned char* src = (o < CH ? kg + o : vg + (o - CH)) + (lane << 4);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds_stage + o), 16, 0, 0);
    }
}


// QK_FMT / V_FMT: QATTN_FMT_E4M3 (0) or QATTN_FMT_E5M2 (1) == the MFMA's cbsz/blgp selector.
template <int D, int QK_FMT, int V_FMT, bool CAUSAL, bool TOKEN>
__global__ __launch_bounds__(kThreads, 2) void attn_fwd_kernel_v1(const AttnParams p) {
    constexpr int CH = 64 * D;          // bytes of one K (or V) chunk
    constexpr int STAGE = 2 * CH;       // K chunk + V chunk
    constexpr int KS = D / 64;          // QK^T k-steps
    constexpr int MB = D / 32;          // O^T row blocks
    constexpr int ROUNDS = 2 * CH / (kThreads * 16);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hh = lane >> 5;

    // ---- block -> (head, query block).  Blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a
    // contiguous range of heads so the 1-2 heads it is working on keep their K/V in its private 4 MiB L2.
    int bid = blockIdx.x;
    int head, qb;
    const int NH = p.B * p.Hq;
    const int nqb_l = p.v1_qb_n > 0 ? p.v1_qb_n : p.nqb;  // this launch covers query blocks [0, nqb_l)
    if (p.xcd_remap) {
        const int xcd = bid & 7, idx = bid >> 3;
        head = xcd * (NH >> 3) + idx / nqb_l;
        qb = idx % nqb_l;
    } else {
        head = bid / nqb_l;
        qb = bid % nqb_l;
    }
    if (CAUSAL) qb = nqb_l - 1 - qb;  // heaviest query blocks first
    const int b = head / p.Hq, h = head % p.Hq;
    const int hkv = h / (p.Hq / p.Hkv);
    const long kv_head = (long)b * p.Hkv + hkv;
    const int q0_wg = qb * kQPerWG;
    const int q0 = q0_wg + wave * kQPerWave;  // first query row of this wave
    const int qrow = q0 + ql;                 // this lane's query row

    const unsigned char* kg = p.k + kv_head * (long)p.nchunks * CH;
    const unsigned char* vg = p.v + kv_head * (long)p.nchunks * CH;

    int nloop = p.nchunks;
    if (CAUSAL) {
        const int last_q = min(q0_wg + kQPerWG, p.Sq) - 1;
        nloop = min(nloop, last_q / 64 + 1);
    }

    // ---- prologue: start the DMA ring, then load Q^T fragments straight to registers
    stage_chunk<D>(kg, vg, smem, wave, lane);
    if (nloop > 1) stage_chunk<D>(kg + CH, vg + CH, smem + STAGE, wave, lane);

    v8i qf[KS];
    {
        const bool qvalid = qrow < p.Sq;
        const unsigned char* qp = p.q + (((long)b * p.Hq + h) * p.Sq + (qvalid ? qrow : 0)) * D + hh * 32;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            v4i lo = *reinterpret_cast<const v4i*>(qp + s * 64);
            v4i hi = *reinterpret_cast<const v4i*>(qp + s * 64 + 16);
            if (!qvalid) { lo = v4i{0, 0, 0, 0}; hi = v4i{0, 0, 0, 0}; }
            qf[s][0] = lo[0]; qf[s][1] = lo[1]; qf[s][2] = lo[2]; qf[s][3] = lo[3];
            qf[s][4] = hi[0]; qf[s][5] = hi[1]; qf[s][6] = hi[2]; qf[s][7] = hi[3];
        }
    }

    // softmax scale in the exp2 domain: c = scale_q * scale_k * sm_scale * log2(e)   (tk/attention.py:204-210)
    float c;
    if (TOKEN) c = p.sm_log2e * (qrow < p.Sq ? p.sq[((long)b * p.Hq + h) * p.Sq + qrow] : 1.0f);
    else c = p.sm_log2e * p.sq[(long)b * p.Hq + h] * p.sk[kv_head];
    const float* skt = TOKEN ? p.sk + kv_head * p.Skv : nullptr;

    v16f o[MB];
#pragma unroll
    for (int m = 0; m < MB; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[m][r] = 0.0f;
    float m_run = -INFINITY;  // running max of the raw (unscaled) scores; TOKEN: of scores already times sk[key]
    float l_run = 0.0f;       // this lane's partial row sum of P'

    // rows that see few keys get the two-term (hi+lo) fp8 P: wave-uniform
    const int visible = CAUSAL ? min(q0 + 1, p.Skv) : p.Skv;
    const bool two_term = visible < kTwoTermKeys;

    const int frag_lane_off = (hh << 10) + (ql << 4);

    for (int c_idx = 0; c_idx < nloop; c_idx++) {
        // chunk c_idx's DMA was issued two iterations ago; at most the next chunk's may stay in flight
        if (c_idx + 1 < nloop) { if (ROUNDS == 1) wait_vmcnt<1>(); else if (ROUNDS == 2) wait_vmcnt<2>(); else wait_vmcnt<4>(); }
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        // every wave is past its reads of stage (c_idx-1)%3 == (c_idx+2)%3: refill it
        if (c_idx + 2 < nloop)
            stage_chunk<D>(kg + (long)(c_idx + 2) * CH, vg + (long)(c_idx + 2) * CH, smem + ((c_idx + 2) % kStages) * STAGE, wave, lane);

        const int k0 = c_idx * 64;
        if (CAUSAL && k0 > q0 + kQPerWave - 1) continue;  // fully masked for this wave (wave-uniform)

        const unsigned char* kbuf = smem + (c_idx % kStages) * STAGE + frag_lane_off;
        const unsigned char* vbuf = kbuf + CH;

        // ---- S^T = K . Q^T   (two 32-key tiles)
        v16f s0, s1;
#pragma unroll
        for (int r = 0; r < 16; r++) { s0[r] = 0.0f; s1[r] = 0.0f; }
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const v8i ka = lds_read_frag(kbuf + ((0 * KS + s) << 11));
            const v8i kb = lds_read_frag(kbuf + ((1 * KS + s) << 11));
            s0 = mfma_f8<QK_FMT, QK_FMT>(ka, qf[s], s0);
            s1 = mfma_f8<QK_FMT, QK_FMT>(kb, qf[s], s1);
        }
        float sc[32];
#pragma unroll
        for (int r = 0; r < 16; r++) { sc[r] = s0[r]; sc[16 + r] = s1[r]; }

        if (TOKEN) {
            // token-wise: score *= scale_k[key]  (inductor/kernels/attention.py:395); key = k0 + 32t + 8j + 4hh + i
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int kk = k0 + 32 * t + 8 * j + 4 * hh;
                    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (kk + 3 < p.Skv) w = *reinterpret_cast<const float4*>(skt + kk);
                    else { if (kk < p.Skv) w.x = skt[kk]; if (kk + 1 < p.Skv) w.y = skt[kk + 1]; if (kk + 2 < p.Skv) w.z = skt[kk + 2]; }
                    sc[16 * t + 4 * j + 0] *= w.x; sc[16 * t + 4 * j + 1] *= w.y;
                    sc[16 * t + 4 * j + 2] *= w.z; sc[16 * t + 4 * j + 3] *= w.w;
                }
        }

        // ---- masking (ragged tail / causal diagonal): wave-uniform branch
        const bool need_mask = (k0 + 64 > p.Skv) || (CAUSAL && k0 + 63 > q0);
        if (need_mask) {
#pragma unroll
            for (int r = 0; r < 32; r++) {
                const int key = k0 + 32 * (r >> 4) + (r & 3) + 8 * ((r & 15) >> 2) + 4 * hh;
                const bool dead = key >= p.Skv || (CAUSAL && key > qrow);
                sc[r] = dead ? -INFINITY : sc[r];
            }
        }

        // ---- row max: 32 in-lane values + the other half-wave's 32
        float mx = sc[0];
#pragma unroll
        for (int r = 1; r < 32; r++) mx = fmaxf(mx, sc[r]);
        {
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        const float m_new = fmaxf(m_run, mx);
        // deferred rescale: only when some row's max grew by more than the threshold (always on the first chunk)
        if (__any((m_new - m_run) * c > kRescaleThr)) {
            const float alpha = (m_new == m_run) ? 1.0f : __builtin_amdgcn_exp2f((m_run - m_new) * c);
#pragma unroll
            for (int m = 0; m < MB; m++)
#pragma unroll
                for (int r = 0; r < 16; r++) o[m][r] *= alpha;
            l_run *= alpha;
            m_run = m_new;
        }
        const float mc = kPShift - m_run * c;

        // ---- P' = exp2(c*s - c*m + shift); partial row sum; e4m3 conversion -> PV B operand
        float pr[32];
#pragma unroll
        for (int r = 0; r < 32; r++) pr[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], c, mc));
        float ls = 0.0f;
#pragma unroll
        for (int r = 0; r < 32; r++) ls += pr[r];
        l_run += ls;

        v8i pb;
#pragma unroll
        for (int w = 0; w < 8; w++) pb[w] = cvt4_fp8<QATTN_FMT_E4M3>(pr[4 * w], pr[4 * w + 1], pr[4 * w + 2], pr[4 * w + 3]);

        // ---- O^T += V^T . P^T
        if (!two_term) {
#pragma unroll
            for (int m = 0; m < MB; m++) {
                const v8i va = lds_read_frag(vbuf + (m << 11));
                o[m] = mfma_f8<V_FMT, QATTN_FMT_E4M3>(va, pb, o[m]);
            }
        } else {
            // residual term: P' - fp8(P') re-quantised to e4m3 (relative error 2^-8 instead of 2^-4)
            v8i pl;
#pragma unroll
            for (int w = 0; w < 8; w++) {
                const float h0 = __builtin_amdgcn_cvt_f32_fp8(pb[w], 0), h1 = __builtin_amdgcn_cvt_f32_fp8(pb[w], 1);
                const float h2 = __builtin_amdgcn_cvt_f32_fp8(pb[w], 2), h3 = __builtin_amdgcn_cvt_f32_fp8(pb[w], 3);
                pl[w] = cvt4_fp8<QATTN_FMT_E4M3>(pr[4 * w] - h0, pr[4 * w + 1] - h1, pr[4 * w + 2] - h2, pr[4 * w + 3] - h3);
            }
#pragma unroll
            for (int m = 0; m < MB; m++) {
                const v8i va = lds_read_frag(vbuf + (m << 11));
                o[m] = mfma_f8<V_FMT, QATTN_FMT_E4M3>(va, pb, o[m]);
                o[m] = mfma_f8<V_FMT, QATTN_FMT_E4M3>(va, pl, o[m]);
            }
        }
    }

    // ---- epilogue: combine the two half-wave partial sums, normalise, convert, store
    float l_tot;
    {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
        l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const float sv = p.sv ? p.sv[kv_head] : 1.0f;
    const float inv = sv / l_tot;
    if (qrow < p.Sq) {
        const long row_off = (((long)b * p.Hq + h) * p.Sq + qrow) * D;
        if (p.out_fmt == QATTN_FMT_BF16) {
            __bf16* op = reinterpret_cast<__bf16*>(p.out) + row_off;
#pragma unroll
            for (int m = 0; m < MB; m++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
                    bf4 t;
#pragma unroll
                    for (int i = 0; i < 4; i++) t[i] = (__bf16)(o[m][4 * j + i] * inv);
                    *reinterpret_cast<bf4*>(op + 32 * m + 8 * j + 4 * hh) = t;
                }
        } else {
            _Float16* op = reinterpret_cast<_Float16*>(p.out) + row_off;
#pragma unroll
            for (int m = 0; m < MB; m++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    h4 t;
#pragma unroll
                    for (int i = 0; i < 4; i++) t[i] = (_Float16)(o[m][4 * j + i] * inv);
                    *reinterpret_cast<h4*>(op + 32 * m + 8 * j + 4 * hh) = t;
                }
        }
        if (p.lse && hh == 0) {
            // ln sum_j exp(score_j) = ln2 * (m*c - shift) + ln(l')
            p.lse[((long)b * p.Hq + h) * p.Sq + qrow] = 0.6931471805599453f * (m_run * c - kPShift) + __logf(l_tot);
        }
    }
}

template <int D, int FMT, bool CAUSAL>
static int launch_attn_v1_t(const AttnParams& p, int scale_mode, hipStream_t st) {
    const int grid = p.B * p.Hq * (p.v1_qb_n > 0 ? p.v1_qb_n : p.nqb);
    const size_t lds = (size_t)kStages * 2 * 64 * D;
    if (scale_mode == QATTN_SCALE_TOKEN) {
        auto kern = attn_fwd_kernel_v1<D, FMT, FMT, CAUSAL, true>;
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return QATTN_ERR_LAUNCH;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, st, p);
    } else {
        auto kern = attn_fwd_kernel_v1<D, FMT, FMT, CAUSAL, false>;
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return QATTN_ERR_LAUNCH;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, st, p);
    }
    return QATTN_OK;
}

template <int D>
static int launch_attn_v1_d(const AttnParams& p, int fmt, int causal, int scale_mode, hipStream_t st) {
    if (fmt == QATTN_FMT_E4M3) return causal ? launch_attn_v1_t<D, QATTN_FMT_E4M3, true>(p, scale_mode, st) : launch_attn_v1_t<D, QATTN_FMT_E4M3, false>(p, scale_mode, st);
    return causal ? launch_attn_v1_t<D, QATTN_FMT_E5M2, true>(p, scale_mode, st) : launch_attn_v1_t<D, QATTN_FMT_E5M2, false>(p, scale_mode, st);
}




int launch_attn_v1(const AttnParams& pin, int D, int fmt, int causal, int scale_mode, hipStream_t st) {
    AttnParams p = pin;
    p.v1_qb_n = 0;
    if (D == 64) return launch_attn_v1_d<64>(p, fmt, causal, scale_mode, st);
    if (D == 128) return launch_attn_v1_d<128>(p, fmt, causal, scale_mode, st);
    return launch_attn_v1_d<256>(p, fmt, causal, scale_mode, st);
}

}  // namespace qattn
