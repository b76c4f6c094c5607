#!/usr/bin/env python3
"""Development: which key chunk does the 4-wave kernel get wrong?  V is zero outside one 64-key chunk at a time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ab
B, H, S, D = 1, 8, 1024, 128
torch.manual_seed(1)
q, k, v0 = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
Ln, Lo = ab.load("quantumattention_amd/libqattn_hip.so"), ab.load("tools/ab_libs/libqattn_r4.so")
for j in list(range(16)):
    v = torch.zeros_like(v0)
    if j >= 0: v[:, :, 64 * j:64 * j + 64] = v0[:, :, 64 * j:64 * j + 64]
    else: v = v0.clone()
    outs = []
    for L in (Ln, Lo):
        x = ab.Variant("x", L, q, k, v, False, 0)
        x.out.fill_(float("nan")); x.attn(ab.PREC["fast"]); torch.cuda.synchronize(); outs.append(x.out.float().clone())
    d = (outs[0] - outs[1]).abs()
    if True:
        rn, ro = outs[0].view(B, H, S // 256, 8, 32, D), outs[1].view(B, H, S // 256, 8, 32, D)
        print("   per 32-row group new/old:", [round((rn[:, :, :, g].norm() / ro[:, :, :, g].norm()).item(), 3) for g in range(8)])
        print("   per block new/old:", [round((rn[:, :, b].norm() / ro[:, :, b].norm()).item(), 3) for b in range(S // 256)])
        print("   per head new/old:", [round((rn[:, h].norm() / ro[:, h].norm()).item(), 3) for h in range(H)])
    print(f"chunk {j:2d}: max diff {d.max().item():.5f}  |old| max {outs[1].abs().max().item():.4f}  new/old norm {(outs[0].norm() / outs[1].norm()).item():.4f}")
