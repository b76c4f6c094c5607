#!/usr/bin/env python3
"""Development: in-kernel clock and KV-sweep cycles per wave of the fused C2 step (stamped instantiation of the attention kernel), for one or
more builds of the library.
   python tools/kernel_clock.py name=path ... [--prec fast]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantumattention_amd import _native
prec = sys.argv[sys.argv.index("--prec") + 1] if "--prec" in sys.argv else "fast"
torch.manual_seed(0)
q, k, v = (torch.randn(4, 32, 4096, 128, dtype=torch.bfloat16, device="cuda") for _ in range(3))
import ctypes
for a in [x for x in sys.argv[1:] if "=" in x]:
    name, path = a.split("=", 1)
    _native._lib = None
    _native.LIB_PATH = os.path.abspath(path)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        clk, cyc, _ = _native.measure_attention_clock(q, k, v, precision=prec, calls=300)
    print(f"{name:10s} clock {clk:.3f} GHz  sweep cycles/wave {cyc:.0f}  = {cyc / 66:.0f} per iteration (66)")
