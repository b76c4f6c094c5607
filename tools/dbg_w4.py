#!/usr/bin/env python3
"""Development: where does the 4-wave kernel differ from the 8-wave one?  error by 32-row group of a block, by 32-column block of D, by block index"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ab
B, H, S, D = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,8,4096,128").split(","))
torch.manual_seed(1)
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
new = ab.Variant("new", ab.load(sys.argv[2] if len(sys.argv) > 2 else "quantumattention_amd/libqattn_hip.so"), q, k, v, False, 0)
old = ab.Variant("old", ab.load("tools/ab_libs/libqattn_r4.so"), q, k, v, False, 0)
for path in ("attn", "fused"):
    outs = []
    for x in (new, old):
        x.out.fill_(float("nan")); getattr(x, path)(ab.PREC["fast"]); torch.cuda.synchronize(); outs.append(x.out.float().clone())
    d = (outs[0] - outs[1]).abs()
    print(path, "max", d.max().item(), "nan", torch.isnan(outs[0]).sum().item())
    dr = d.view(B, H, S // 256, 8, 32, 4, 32)
    print(" by 32-row group:", [round(x, 4) for x in dr.amax(dim=(0, 1, 2, 4, 5, 6)).tolist()])
    print(" by d block     :", [round(x, 4) for x in dr.amax(dim=(0, 1, 2, 3, 4, 6)).tolist()])
    print(" by block       :", [round(x, 4) for x in dr.amax(dim=(0, 1, 3, 4, 5, 6)).tolist()])
    print(" by head        :", [round(x, 4) for x in dr.amax(dim=(0, 2, 3, 4, 5, 6)).tolist()])
    print(" by row in group:", [round(x, 3) for x in dr.amax(dim=(0, 1, 2, 3, 5, 6)).tolist()])
    print(" by col in dblk :", [round(x, 3) for x in dr.amax(dim=(0, 1, 2, 3, 4, 5)).tolist()])
    r = (outs[0].norm() / outs[1].norm()).item()
    print(" norm ratio", r, " mean rel err", (d.mean() / outs[1].abs().mean()).item())
