#!/usr/bin/env python3
"""Numpy/torch model of the HIP attention kernel's numerics (chunked online softmax, fp8 P, deferred rescale) used
to budget accuracy before changing the kernel.  Test infrastructure only (uses the oracle as the reference).

  python tools/models/sim_kernel.py
"""
import math
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # the repository root
import oracle  # noqa: E402

E4M3_LUT = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).float()  # byte -> value


def fp8_round(x):
    return x.to(torch.float8_e4m3fn).float()


def sim_head(q, k, v, c, sv, causal, mode, shift=5.0, thr=3.0, bias=0.0, two_term=False, chunk=64):
    """q [Sq,D], k,v [Skv,D] float32 (already de-quantised payload values, scales folded into c / sv).
    mode: 'exact' (v_exp + RNE fp8, l from unrounded p) | 'exact_lq' (l from rounded p) | 'byte' (Schraudolph byte)"""
    Sq, D = q.shape
    Skv = k.shape[0]
    s_all = (q.double() @ k.double().T).float()  # raw scores (MFMA fp32 accumulate ~ exact for fp8 inputs)
    m_run = torch.full((Sq,), -1e30)
    l_run = torch.zeros(Sq)
    o = torch.zeros(Sq, D, dtype=torch.float64)
    qi = torch.arange(Sq)[:, None]
    for k0 in range(0, Skv, chunk):
        s = s_all[:, k0:k0 + chunk].clone()
        kj = torch.arange(k0, min(k0 + chunk, Skv))[None, :]
        if causal:
            s = torch.where(kj > qi, torch.tensor(-float("inf")), s)
        mx = s.max(dim=1).values
        # deferred rescale in 32-row wave groups
        need = ((mx - m_run) * c > thr).view(-1, 32).any(dim=1).repeat_interleave(32)
        m_new = torch.where(need, torch.maximum(m_run, mx), m_run)
        alpha = torch.exp2((m_run - m_new) * c)
        o *= alpha[:, None].double()
        l_run = l_run * alpha
        m_run = m_new
        x = s * c + (shift - m_run * c)[:, None]
        if mode == "byte":
            b = torch.clamp(torch.round(8.0 * x + 56.0 + bias), 0, 126)  # torch.round = RNE
            b = torch.where(torch.isnan(b), torch.zeros_like(b), b).to(torch.uint8)
            ph = E4M3_LUT[b.long()]
            l_run = l_run + ph.sum(dim=1)
            pterms = [ph]
        else:
            p = torch.exp2(x)
            ph = fp8_round(p)
            pterms = [ph]
            if two_term:
                pterms.append(fp8_round(p - ph))
            if mode == "exact":
                l_run = l_run + p.sum(dim=1)
            else:
                l_run = l_run + sum(t.sum(dim=1) for t in pterms)
        vv = v[k0:k0 + chunk].double()
        for t in pterms:
            o += t.double() @ vv
    out = (o * (sv / l_run.double())[:, None]).float()
    return out.to(torch.bfloat16).float()


def run(name, S, D, causal, seed=0, heads=2, rows=None):
    torch.manual_seed(seed)
    q = torch.randn(1, heads, S, D, dtype=torch.bfloat16)
    k = torch.randn(1, heads, S, D, dtype=torch.bfloat16)
    v = torch.randn(1, heads, S, D, dtype=torch.bfloat16)
    b16 = lambda t: t.contiguous().view(torch.int16).numpy().view(np.uint16)
    q8, sq = oracle.quantize_fp8(b16(q), oracle.FMT_BF16, "head")
    k8, sk = oracle.quantize_fp8(b16(k), oracle.FMT_BF16, "head")
    v8, sv = oracle.quantize_fp8(b16(v), oracle.FMT_BF16, "head")
    ref = oracle.attention_forward(q8, k8, v8, 0, 0, 0, sq, sk, sv, causal=causal)
    res = {}
    for label, kw in [
        ("exact(l=fp32 p)", dict(mode="exact")),
        ("exact(l=fp8 p)", dict(mode="exact_lq")),
        ("byte bias0", dict(mode="byte", bias=0.0)),
        ("byte bias-.25", dict(mode="byte", bias=-0.25)),
        ("byte bias-.35", dict(mode="byte", bias=-0.35)),
        ("byte bias-.45", dict(mode="byte", bias=-0.45)),
        ("byte b-.35 sh7 thr1", dict(mode="byte", bias=-0.35, shift=7.0, thr=1.0)),
    ]:
        errs, rms = [], []
        for h in range(heads):
            qf = torch.from_numpy(oracle.fp8_to_f32(q8[0, h]))
            kf = torch.from_numpy(oracle.fp8_to_f32(k8[0, h]))
            vf = torch.from_numpy(oracle.fp8_to_f32(v8[0, h]))
            c = float(sq[0, h]) * float(sk[0, h]) * (1.0 / math.sqrt(D)) * 1.4426950408889634
            out = sim_head(qf, kf, vf, c, float(sv[0, h]), causal, **kw).numpy()
            d = np.abs(out - ref[0, h])
            if rows is not None:
                d = d[rows]
            errs.append(d.max()); rms.append(np.sqrt((d ** 2).mean()))
        res[label] = (max(errs), float(np.mean(rms)))
    print(f"{name}: " + " | ".join(f"{k_}: max {a:.4f} rms {b:.5f}" for k_, (a, b) in res.items()))


if __name__ == "__main__":
    run("S4096 D128 full  ", 4096, 128, False, heads=1)
    run("S4096 D128 causal rows>=1024", 4096, 128, True, heads=1, rows=slice(1024, None))
    run("S1024 D128 full  ", 1024, 128, False)
    run("S2048 D128 causal rows>=1024", 2048, 128, True, rows=slice(1024, None))
