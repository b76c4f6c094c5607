#!/usr/bin/env python3
"""Rows whose weight sits on K SIMILAR keys: R = 1 / w_max only bounds the largest weight, the one-term error is
statistical, ~ eps_rms * sqrt(sum w^2) * |v - O| -> the second statistic is the effective key count N_eff = l^2 / sum P'^2.
The kernel gets sum P'^2 from the matrix pipe: the e4m3 byte of P' read as an e5m2 number is 0.444 .. 0.5 of P'^2 (the
exponent field weighs twice as much), so one more row-sum MFMA with the B format switched to bf8 accumulates it.

NumPy/torch model (test infrastructure; uses the oracle): the construction of tests/test_gpu_precision.py, the one-term
byte-exponential kernel arithmetic of tools/models/sim_kernel.py, both statistics, and the worst error among the rows each rule accepts.

  python tools/models/sim_heavy.py
"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # the repository root
import oracle  # noqa: E402

E4M3_LUT = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).float()
E5M2_LUT = torch.arange(256, dtype=torch.uint8).view(torch.float8_e5m2).float()


def heavy_inputs(S, D, K, gap, jitter, late, seed, H=1):
    """q, k, v with K keys that carry every row: score gap `gap` nats above the N(0,1) background, `jitter` nats of spread among
    the heavy keys.  late: the heavy keys sit in the last 256 positions only."""
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(1, H, S, D, generator=g)
    k = torch.randn(1, H, S, D, generator=g)
    v = torch.randn(1, H, S, D, generator=g)
    u = torch.randn(D, generator=g)
    u /= u.norm()
    a = math.sqrt(gap * math.sqrt(D))
    idx = (S - 256 + torch.randperm(256, generator=g)[:K]) if late else torch.randperm(S, generator=g)[:K]
    # background keys lose their component along u (their scores stay N(0,1)); heavy keys = a u + a small random part
    k = k - (k @ u)[..., None] * u
    q = q - (q @ u)[..., None] * u + a * u
    kh = torch.randn(1, H, K, D, generator=g)
    kh = kh - (kh @ u)[..., None] * u
    k[:, :, idx] = jitter * kh + a * u
    return q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16), idx


def sim_one_term(q, k, v, c, causal, bias=-0.3, shift=5.0, thr=3.0, chunk=64):
    Sq, D = q.shape
    Skv = k.shape[0]
    s_all = (q.double() @ k.double().T).float()
    m_run = torch.full((Sq,), -1e30)
    m_true = torch.full((Sq,), -1e30)
    l = torch.zeros(Sq)
    q2 = torch.zeros(Sq)     # sum of the bytes read as e5m2 (the extra MFMA)
    p2 = torch.zeros(Sq)     # true sum P'^2
    o = torch.zeros(Sq, D, dtype=torch.float64)
    qi = torch.arange(Sq)[:, None]
    for k0 in range(0, Skv, chunk):
        s = s_all[:, k0:k0 + chunk].clone()
        if causal:
            kj = torch.arange(k0, min(k0 + chunk, Skv))[None, :]
            s = torch.where(kj > qi, torch.tensor(-float("inf")), s)
        mx = s.max(dim=1).values
        m_true = torch.maximum(m_true, mx)
        need = ((mx - m_run) * c > thr).view(-1, 32).any(dim=1).repeat_interleave(32)
        m_new = torch.where(need, torch.maximum(m_run, mx), m_run)
        alpha = torch.exp2((m_run - m_new) * c)
        o *= alpha[:, None].double(); l = l * alpha; q2 = q2 * alpha * alpha; p2 = p2 * alpha * alpha
        m_run = m_new
        x = s * c + (shift - m_run * c)[:, None]
        b = torch.clamp(torch.round(8.0 * x + 56.0 + bias), 0, 126)
        b = torch.where(torch.isnan(b), torch.zeros_like(b), b).long()
        ph = E4M3_LUT[b]
        l = l + ph.sum(1); q2 = q2 + E5M2_LUT[b].sum(1); p2 = p2 + (ph * ph).sum(1)
        o += ph.double() @ v[k0:k0 + chunk].double()
    pmax = torch.exp2(shift + (m_true - m_run) * c)
    out = (o / l.double()[:, None]).float().to(torch.bfloat16).float()
    return out, (l / pmax).numpy(), (l * l / p2).numpy(), (l * l / q2).numpy()


def run(S, D, K, gap, jitter, late, causal=False, seed=0):
    q, k, v, idx = heavy_inputs(S, D, K, gap, jitter, late, seed)
    b16 = lambda t: t.contiguous().view(torch.int16).numpy().view(np.uint16)
    q8, sq = oracle.quantize_fp8(b16(q), oracle.FMT_BF16, "head")
    k8, sk = oracle.quantize_fp8(b16(k), oracle.FMT_BF16, "head")
    _, _, vdq = oracle.quantize_v_block(b16(v), oracle.FMT_BF16)
    ref = oracle.attention_forward(q8, k8, vdq, 0, 0, oracle.FMT_BF16, sq, sk, None, causal=causal)[0, 0]
    qf = torch.from_numpy(oracle.fp8_to_f32(q8[0, 0])); kf = torch.from_numpy(oracle.fp8_to_f32(k8[0, 0]))
    vf = torch.from_numpy(oracle.bf16_bits_to_f32(vdq[0, 0]))
    c = float(sq[0, 0]) * float(sk[0, 0]) / math.sqrt(D) * 1.4426950408889634
    out, R, neff, neff5 = sim_one_term(qf, kf, vf, c, causal)
    err = np.abs(out.numpy() - ref).max(axis=1)
    # weight the K keys hold (fp64 softmax on the quantised operands)
    sc = (qf.double() @ kf.double().T) * (c / 1.4426950408889634)
    if causal:
        sc = sc.masked_fill(torch.arange(S)[None, :] > torch.arange(S)[:, None], -float("inf"))
    w = torch.softmax(sc, dim=1)
    held = w[:, idx].sum(1).numpy()
    rows = slice(1024, None) if causal else slice(None)
    err, R, neff, neff5, held = err[rows], R[rows], neff[rows], neff5[rows], held[rows]
    ok24 = R >= 24
    line = (f"K={K:3d} gap {gap:4.1f} jit {jitter:.2f} {'late' if late else 'unif'} {'causal' if causal else '      '}: held med {np.median(held):.3f} | "
            f"R med {np.median(R):6.1f} frac R<24 {np.mean(~ok24):.3f} | Neff med {np.median(neff):6.1f} (e5m2 est x0.472: {np.median(neff5) * 0.472:6.1f}) | "
            f"max err all {err.max():.4f}, rows R>=24 {err[ok24].max() if ok24.any() else 0:.4f}")
    for T in (96, 128, 160, 192, 256):
        acc = ok24 & (neff5 * 0.472 >= T)
        line += f" | T{T}: {err[acc].max() if acc.any() else 0:.4f} ({np.mean(acc):.2f})"
    print(line, flush=True)


if __name__ == "__main__":
    S, D = 4096, 128
    for late in (False, True):
        for K, jit in ((8, 0.15), (16, 0.15), (32, 0.15), (40, 0.15), (48, 0.2), (64, 0.2), (100, 0.3), (160, 0.3), (256, 0.3)):
            run(S, D, K, 9.0, jit, late)
    run(S, D, 48, 9.0, 0.2, False, causal=True)
    # flat data: what the rule costs (rows refused among N(0,1) rows)
    torch.manual_seed(0)
    for S_ in (1024, 4096):
        q, k, v = (torch.randn(S_, D) for _ in range(3))
        c = 1.4426950408889634 / math.sqrt(D)
        out, R, neff, neff5 = sim_one_term(q, k, v, c, False)
        print(f"flat S={S_}: R min {R.min():.1f} frac<24 {np.mean(R < 24):.5f} | Neff min {neff.min():.0f} med {np.median(neff):.0f} | e5m2 est min {neff5.min() * 0.472:.0f}"
              f" | frac est<192 {np.mean(neff5 * 0.472 < 192):.5f} <256 {np.mean(neff5 * 0.472 < 256):.5f}")
