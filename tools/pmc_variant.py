#!/usr/bin/env python3
"""Development: N back-to-back attention launches (pre-quantised operands) of ONE build of the library, to be run under rocprofv3 --pmc.
   LIB=<path> PREC=fast|auto SHAPE=B,H,S,D CALLS=n python3 tools/pmc_variant.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ab
B, H, S, D = (int(x) for x in os.environ.get("SHAPE", "4,32,4096,128").split(","))
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
x = ab.Variant("x", ab.load(os.environ["LIB"]), q, k, v, False, 0)
for _ in range(int(os.environ.get("CALLS", "20"))):
    x.attn(ab.PREC[os.environ.get("PREC", "fast")])
torch.cuda.synchronize()
