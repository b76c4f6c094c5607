#!/usr/bin/env python3
"""Randomised parity sweep (test infrastructure: uses the oracle): N random configurations of the fused step and of the separate
C calls -- B, Hq / Hkv (GQA), Sq, Skv (ragged, Sq != Skv where not causal), D, causal, fp8 format, scaling, 16-bit dtype,
precision, score spread (q x 1 .. x 3), one-outlier rows -- each checked against the fp64 oracle on the same quantised inputs
(quantiser bit-exact, attention max-abs < 2^-6 max(1, |O| / 2, std V) for AUTO / ACCURATE).  Sizes the oracle finishes in a second or two.

  python tools/fuzz_parity.py [N=120] [seed=0] [adv]     prints one line per case, a summary, exit status 1 on any failure
  adv: two cases of three get an adversarial score / value structure (function adversarial below)
"""
import os, sys, time
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
import quantumattention_amd as qa  # noqa: E402
from quantumattention_amd import _native  # noqa: E402
from tests.gpu_utils import (FMT, TDT, bits16, bits8, check_path_structure, err_stats, fmt16, fused_call, fused_step_uses_block_v, oracle_for_fp8_path,  # noqa: E402
                             out_to_f32, unpack_frag)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ADV = len(sys.argv) > 3 and sys.argv[3] == "adv"      # round 5: adversarial score structures on two cases of three (below)


def adversarial(q, k, v, rng, causal):
    """Round 5 (VERDICT r4 Weak-3: grow the generators, not the count).  Rewrites q / k / v in place with a structure aimed at one rule of the
    precision machinery and returns its name.  A planted key j* gets the score beta above a row's N(0, 1) rest by
    q_r += beta k_j* / |k_j*|^2 sqrt(D) (head 0 of the row's KV group): its weight is ~ e^beta / (n e^0.5 + e^beta)."""
    B, Hq, Sq, D = q.shape
    Hkv, Skv = k.shape[1], k.shape[2]
    grp = Hq // Hkv
    kind = str(rng.choice(["mild_and_severe_rows", "a_few_equal_keys", "about_96_peaked_rows", "bimodal_scores", "late_dominant_key",
                           "heavy_tailed_v", "zero_rows", "large_magnitudes"]))

    def plant(rows, betas, keys_of_row):
        for r, beta, js in zip(rows, betas, keys_of_row):
            for j in js:
                if causal and j > r + (Skv - Sq):
                    j = int(rng.integers(0, max(1, min(Skv, r + 1))))
                for h in range(Hq):
                    kj = k[:, h // grp, j].float()
                    q[:, h, r] = (q[:, h, r].float() + beta * kj / (kj * kj).sum(-1, keepdim=True).clamp_min(1e-6) * D ** 0.5).to(q.dtype)

    n_flat = max(Skv, 2) * 1.65
    if kind == "mild_and_severe_rows":      # weights from 1/30 to 0.6: both sides of R = 24 (flag) and R = 8 (16-bit-V rescue), scattered over the blocks
        rows = rng.choice(Sq, size=min(Sq, int(rng.integers(1, 60))), replace=False)
        w = np.exp(rng.uniform(np.log(1 / 30), np.log(0.6), size=len(rows)))
        plant(rows, np.log(w / (1 - w) * n_flat), [[int(rng.integers(Skv))] for _ in rows])
    elif kind == "a_few_equal_keys":        # 2 .. 6 keys of weight ~ 1/30 .. 1/12 each: between the R test and the effective key count
        rows = rng.choice(Sq, size=min(Sq, int(rng.integers(1, 40))), replace=False)
        w = rng.uniform(1 / 30, 1 / 12, size=len(rows))
        plant(rows, np.log(w / (1 - 6 * w).clip(0.3) * n_flat), [list(rng.choice(Skv, size=min(Skv, int(rng.integers(2, 7))), replace=False)) for _ in rows])
    elif kind == "about_96_peaked_rows":    # kMaxRescueRows = 96 per 256-row block: 80 .. 112 sharp rows in block 0 (rescue on one side, block repeat on the other)
        n = min(Sq, int(rng.integers(80, 113)))
        rows = rng.choice(min(Sq, 256), size=min(n, min(Sq, 256)), replace=False) + (256 * int(rng.integers(0, max(1, Sq // 256))) if Sq >= 512 else 0)
        rows = rows[rows < Sq]
        plant(rows, np.full(len(rows), np.log(0.3 / 0.7 * n_flat)), [[int(rng.integers(Skv))] for _ in rows])
    elif kind == "bimodal_scores":          # half of the keys sit 2 .. 4 nats above the other half for every row (no single heavy key, a short effective key count)
        shift = float(rng.uniform(2.0, 4.0))
        u = torch.randn(B, Hkv, 1, D)
        u = u / u.norm(dim=-1, keepdim=True)
        sel = torch.from_numpy(rng.integers(0, 2, size=Skv).astype(np.float32))[None, None, :, None]
        k.copy_((k.float() + sel * u * shift ** 0.5 * D ** 0.25).to(k.dtype))
        q.copy_((q.float() + u.repeat_interleave(grp, 1) * shift ** 0.5 * D ** 0.25).to(q.dtype))
    elif kind == "late_dominant_key":       # the key that carries the row sits in the LAST chunk (after the first-chunk forecast and most of the statistics)
        rows = rng.choice(Sq, size=min(Sq, int(rng.integers(1, 200))), replace=False)
        plant(rows, np.full(len(rows), np.log(0.5 * n_flat)), [[Skv - 1 - int(rng.integers(0, min(Skv, 64)))] for _ in rows])
    elif kind == "heavy_tailed_v":          # V = x^3: a few entries carry a chunk's abs-max (the block-scaled fp8 V at its worst)
        v.copy_((v.float() ** 3).clamp(-6.0e4, 6.0e4).to(v.dtype))   # (finite in fp16)
    elif kind == "zero_rows":               # all-equal scores: zero query rows, zero keys
        q[:, :, rng.choice(Sq, size=min(Sq, 5), replace=False)] = 0
        k[:, :, rng.choice(Skv, size=min(Skv, 5), replace=False)] = 0
    else:                                   # large_magnitudes: inputs near the top of fp16's range (scores re-normalised by a small q), bf16 alike
        k.copy_((k.float() * 2000.0).to(k.dtype))
        q.copy_((q.float() / 2000.0).to(q.dtype))
        v.copy_((v.float() * 500.0).clamp(-6.0e4, 6.0e4).to(v.dtype))
    return kind
rng = np.random.default_rng(seed)
TOL = 2.0 ** -6
fails = 0
worst = {}
t0 = time.time()
for case in range(N):
    D = int(rng.choice([64, 128, 128, 256]))
    causal = bool(rng.integers(2))
    Hkv = int(rng.choice([1, 2, 3, 4]))
    Hq = Hkv * int(rng.choice([1, 1, 2, 4]))
    B = int(rng.choice([1, 1, 2, 3]))
    Skv = int(rng.choice([rng.integers(1, 130), rng.integers(130, 700), rng.integers(700, 1500), rng.integers(1500, 2600)]))
    Sq = Skv if (causal or rng.integers(2)) else int(rng.integers(1, 1200))
    while B * Hq * Sq * Skv > 3.0e7:       # keep the fp64 oracle to a couple of seconds
        Skv = max(1, Skv // 2); Sq = Skv if causal else max(1, Sq // 2)
    fp8 = str(rng.choice(["e4m3", "e4m3", "e5m2"]))
    scaling = str(rng.choice(["head-wise", "head-wise", "token-wise"]))
    dtype = torch.bfloat16 if rng.integers(4) else torch.float16
    precision = str(rng.choice(["auto", "auto", "accurate", "fast"]))
    spread = float(rng.choice([1.0, 1.0, 1.5, 2.0, 3.0]))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    q = (torch.randn(B, Hq, Sq, D, generator=g) * spread).to(dtype)
    k = torch.randn(B, Hkv, Skv, D, generator=g).to(dtype)
    v = (torch.randn(B, Hkv, Skv, D, generator=g) * float(rng.choice([1.0, 0.05, 30.0]))).to(dtype)
    if rng.integers(3) == 0 and Skv > 8:    # one outlier key per head
        k[:, :, int(rng.integers(Skv))] *= 3.0
    structure = adversarial(q, k, v, rng, causal) if (ADV and rng.integers(3) > 0) else "plain"
    m = "head" if scaling == "head-wise" else "token"
    q8, sq = oracle.quantize_fp8(bits16(q), fmt16(dtype), m, FMT[fp8])
    k8, sk = oracle.quantize_fp8(bits16(k), fmt16(dtype), m, FMT[fp8])
    vb = fused_step_uses_block_v(D, scaling, dtype, Skv)
    ref = oracle_for_fp8_path(q8, k8, bits16(v), sq, sk, fp8=fp8, v_dtype=dtype, scaling=m, causal=causal, v_block=vb, fused=True)
    ref_sep = oracle_for_fp8_path(q8, k8, bits16(v), sq, sk, fp8=fp8, v_dtype=dtype, scaling=m, causal=causal)   # (the separate calls: fp8 V with one scale per head, on every row)
    qc, kc, vc = q.cuda(), k.cuda(), v.cuda()
    views = bool(rng.integers(2))
    if views:   # the same values as transposed views of [B,S,H,D] tensors (include/qattn_strided.h): the strided addressing of every kernel, and
        #         half of the time the output in the query's layout, against the oracle
        qc, kc, vc = (t.transpose(1, 2).contiguous().transpose(1, 2) for t in (qc, kc, vc))
    kw_views = {"output_layout": "like_query"} if (views and rng.integers(2)) else {}
    # the fused entry with its row_path: every row is graded against THE oracle of the path the kernel reports (tests/gpu_utils.py)
    fused, path = fused_call(qc, kc, vc, causal=causal, precision=precision, fp8=fp8, scaling=scaling, **kw_views)
    check_path_structure(path, Sq, Skv, causal, precision, D == 128 and scaling == "head-wise")
    if case % 8 == 0:   # ... and the public interface is that call with row_path = NULL
        with qa.config.patch({"attention.precision": precision, "attention.fp8_format": fp8}):
            fn = qa.fp8_attn_func if scaling == "head-wise" else qa.fp8_token_wise_attn_func
            assert np.array_equal(out_to_f32(fn(qc, kc, vc, is_causal=causal)), fused)
    qg8, sqg = _native.quant_fp8(qc, scaling=scaling, fp8_dtype=TDT[fp8])
    kf, skg = _native.quant_fp8(kc, scaling=scaling, fp8_dtype=TDT[fp8], layout=_native.LAYOUT_KFRAG)
    vf, svg = _native.quant_fp8(vc, scaling="head-wise", fp8_dtype=TDT[fp8], layout=_native.LAYOUT_VFRAG)
    sep = out_to_f32(_native.fp8_attention_forward(qg8, kf, vf, sqg, skg, svg, Hkv=Hkv, Skv=Skv, out_dtype=dtype, is_causal=causal,
                                                    scaling=scaling, precision=precision))
    # the 16-bit sibling path on the same tensors (fp64 SDPA of the 16-bit inputs; exact exponentials: 2^-7)
    mx16 = tol16 = 0.0
    if case % 2 == 0:
        f16 = fmt16(dtype)
        ref16 = oracle.attention_forward(bits16(q), bits16(k), bits16(v), f16, f16, f16, causal=causal)
        got16 = out_to_f32(qa.attn_func(qc, kc, vc, is_causal=causal))
        mx16, _ = err_stats(got16, ref16)
        # (a bf16 output near |O| = 100 rounds by up to 0.25 on its own: the bound follows the largest output, un-halved)
        tol16 = 2.0 ** -7 * max(1.0, float(np.abs(ref16).max()), float(v.float().std()))
        if not (np.isfinite(got16).all() and mx16 < tol16):
            fails += 1
            print(f"FAIL #{case:3d} 16-bit attn_func: {mx16:.4f} (tol {tol16:.4f})", flush=True)
    # the 16-bit-V mode of the fp8 entry (csrc/qattn_pv16.h: the reference kernel's own P.V numerics): fp64 SDPA of the same
    # quantised q, k with the ORIGINAL 16-bit V; P carries 8 (bf16) / 11 (fp16) mantissa bits: 2^-7 max(1, |O|, std V)
    mxv = tolv = 0.0
    if case % 2 == 1:
        refv = oracle.attention_forward(q8, k8, bits16(v), FMT[fp8], FMT[fp8], fmt16(dtype), sq, sk, None, scale_mode=m, causal=causal)
        gotv = out_to_f32(_native.fp8_attention_forward(qg8, kf, vc, sqg, skg, None, Hkv=Hkv, Skv=Skv, out_dtype=dtype, is_causal=causal,
                                                        scaling=scaling))
        mxv, _ = err_stats(gotv, refv)
        tolv = 2.0 ** -7 * max(1.0, float(np.abs(refv).max()), float(v.float().std()))
        if not (np.isfinite(gotv).all() and mxv < tolv):
            fails += 1
            print(f"FAIL #{case:3d} 16-bit-V mode: {mxv:.4f} (tol {tolv:.4f})", flush=True)
    # quantiser: bit-exact payloads and scales
    q_ok = np.array_equal(bits8(qg8), q8) and np.array_equal(sqg.cpu().numpy(), sq) and \
        np.array_equal(unpack_frag(bits8(kf), _native.LAYOUT_KFRAG, B, Hkv, Skv, D)[:, :, :Skv], k8) and np.array_equal(skg.cpu().numpy(), sk)
    # the bound is absolute for N(0,1)-like V (errors are ~ eps w |v - O|): it scales with V's spread, and with |O| for the output rounding
    tol = TOL * max(1.0, float(np.abs(ref.fp8v).max()) / 2, float(v.float().std()))
    if structure == "heavy_tailed_v":
        # the bound is 0.074 w |v - O| with w < 1 / 24 for the keys a one-term row may keep: 2^-6 for |v - O| up to
        # ~ 4.5, what N(0, 1) values reach -- x^3 values reach 15 .. 30 standard deviations, and the bound follows the largest |v|, not the
        # spread (found by this generator, seed 82 case 116: 0.089 against 2^-6 x 3.85 on a row with a weight of 1 / 50 on a |v| of 60)
        tol = max(tol, TOL * float(v.float().abs().max()) / 4.5)
    mx_f, _ = err_stats(fused, ref, path)
    mx_s, _ = err_stats(sep, ref_sep)
    finite = bool(np.isfinite(fused).all() and np.isfinite(sep).all())
    graded = precision != "fast"       # FAST has no bound on peaked rows / rows that see few keys
    ok = q_ok and finite and (not graded or (mx_f < tol and mx_s < tol))
    fails += not ok
    key = (D, scaling, precision)
    worst[key] = max(worst.get(key, 0.0), mx_f / tol, mx_s / tol) if graded else worst.get(key, 0.0)
    print(f"{'ok  ' if ok else 'FAIL'} #{case:3d} B{B} Hq{Hq} Hkv{Hkv} Sq{Sq} Skv{Skv} D{D} {'causal' if causal else 'full  '} {fp8} {scaling[:5]} "
          f"{'bf16' if dtype == torch.bfloat16 else 'fp16'} {precision:8s} q x{spread} {structure}: quant {'exact' if q_ok else 'DIFFERS'} | fused {mx_f:.4f} sep {mx_s:.4f} (tol {tol:.4f})"
          f"{'' if finite else ' NON-FINITE'}{f' | 16-bit {mx16:.4f} (tol {tol16:.4f})' if tol16 else ''}{f' | 16-bit-V {mxv:.4f} (tol {tolv:.4f})' if tolv else ''}", flush=True)
print(f"{N} cases, {fails} failures, {time.time() - t0:.0f} s; worst error / tolerance per (D, scaling, precision):")
for key in sorted(worst):
    print("  ", key, f"{worst[key]:.2f}")
sys.exit(1 if fails else 0)
