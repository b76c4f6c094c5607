#!/usr/bin/env python3
"""Accuracy model for the one-term fp8-P path: per-row error vs the statistics a kernel could flag rows with.
Test infrastructure (uses the oracle).   python tools/models/sim_flag.py"""
import math, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # the repository root
import oracle  # noqa: E402
from tools.models.sim_kernel import sim_head  # noqa: E402

b16 = lambda t: t.contiguous().view(torch.int16).numpy().view(np.uint16)


def case(name, q, k, v, causal=False, mode="byte", bias=-0.3):
    q8, sq = oracle.quantize_fp8(b16(q), oracle.FMT_BF16, "head")
    k8, sk = oracle.quantize_fp8(b16(k), oracle.FMT_BF16, "head")
    v8, sv = oracle.quantize_fp8(b16(v), oracle.FMT_BF16, "head")
    ref = oracle.attention_forward(q8, k8, v8, 0, 0, 0, sq, sk, sv, causal=causal)[0, 0]
    qf = torch.from_numpy(oracle.fp8_to_f32(q8[0, 0])); kf = torch.from_numpy(oracle.fp8_to_f32(k8[0, 0])); vf = torch.from_numpy(oracle.fp8_to_f32(v8[0, 0]))
    D = q.shape[-1]
    c = float(sq[0, 0]) * float(sk[0, 0]) / math.sqrt(D) * 1.4426950408889634
    out1 = sim_head(qf, kf, vf, c, float(sv[0, 0]), causal, mode=mode, bias=bias).numpy()
    out2 = sim_head(qf, kf, vf, c, float(sv[0, 0]), causal, mode="exact", two_term=True).numpy()
    e1 = np.abs(out1 - ref).max(axis=1); e2 = np.abs(out2 - ref).max(axis=1)
    # row statistics on the exact softmax
    s = (qf.double() @ kf.double().T) * (c / 1.4426950408889634)
    if causal:
        S = s.shape[0]; s = s.masked_fill(torch.arange(S)[None, :] > torch.arange(S)[:, None], -float("inf"))
    w = torch.softmax(s, dim=1)
    R = (1.0 / w.max(dim=1).values).numpy()
    neff = (1.0 / (w * w).sum(dim=1)).numpy()
    print(f"{name:34s} one-term max {e1.max():.4f} two-term max {e2.max():.4f} | R: min {R.min():7.1f} med {np.median(R):7.1f} | Neff: min {neff.min():7.1f} med {np.median(neff):7.1f}")
    return e1, e2, R, neff


def main():
    torch.manual_seed(0)
    S, D = 4096, 128
    allr = []
    for sc in (1.0, 1.25, 1.5, 2.0, 3.0, 5.0):
        q = torch.randn(1, 1, S, D, dtype=torch.bfloat16) * sc
        k = torch.randn(1, 1, S, D, dtype=torch.bfloat16); v = torch.randn(1, 1, S, D, dtype=torch.bfloat16)
        allr.append(case(f"S4096 q x{sc}", q, k, v))
    # per-row mixed sharpness
    q = torch.randn(1, 1, S, D, dtype=torch.bfloat16) * torch.linspace(0.5, 6.0, S).view(1, 1, S, 1).to(torch.bfloat16)
    allr.append(case("S4096 mixed sharpness 0.5..6", q, k, v))
    # K equal heavy keys per row (adversarial for a max-weight statistic)
    for K in (16, 48, 128, 256):
        q = torch.randn(1, 1, S, D, dtype=torch.bfloat16); k = torch.randn(1, 1, S, D, dtype=torch.bfloat16)
        idx = torch.randperm(S)[:K]
        k[0, 0, idx] = (torch.randn(K, D) * 0.3 + q[0, 0, :64].float().mean(0) * 0).to(torch.bfloat16)
        # heavy keys: aligned with a common direction every query shares
        u = torch.randn(D); u /= u.norm()
        q = (q.float() + 6.0 * u).to(torch.bfloat16)
        k[0, 0, idx] = (k[0, 0, idx].float() + 6.0 * u).to(torch.bfloat16)
        allr.append(case(f"S4096 {K} heavy keys", q, k, v))
    for sc in (1.0, 2.0, 3.0):
        q = torch.randn(1, 1, 1024, D, dtype=torch.bfloat16) * sc
        k = torch.randn(1, 1, 1024, D, dtype=torch.bfloat16); v = torch.randn(1, 1, 1024, D, dtype=torch.bfloat16)
        allr.append(case(f"S1024 q x{sc}", q, k, v))
    e1 = np.concatenate([a[0] for a in allr]); R = np.concatenate([a[2] for a in allr]); ne = np.concatenate([a[3] for a in allr])
    print("\nworst one-term row error among rows with statistic >= threshold:")
    for thr in (16, 24, 32, 48, 64, 96, 128, 192, 256):
        mR = R >= thr; mN = ne >= thr
        print(f"  thr {thr:4d}: R-rule keeps {mR.mean()*100:5.1f}% worst {e1[mR].max() if mR.any() else 0:.4f} | Neff-rule keeps {mN.mean()*100:5.1f}% worst {e1[mN].max() if mN.any() else 0:.4f}")


if __name__ == "__main__":
    main()
