#!/usr/bin/env python3
"""Development: A/B timing of several builds of libqattn_hip.so in ONE process, interleaved rounds (cdna_hip_programming.md
section 5.4 rule 24: never rank builds by separate invocations).  Every library is loaded through ctypes on its own, quantises
its own operands, and each (library, precision, path) variant is timed with HIP events on the current stream.

  python tools/ab.py [name=path ...] [--shape B,H,S,D] [--causal] [--prec auto,fast] [--paths fused,attn,quant]
                     [--rounds 7] [--calls 20] [--fp8 e4m3] [--scale 1.0] [--settle 0.5]
  paths: fused = qattn_fp8_quant_attention_forward (the bench step), attn = qattn_fp8_attention_forward on pre-quantised
         operands, quant = qattn_quant_qkv_fp8.
  default libraries: new=quantumattention_amd/libqattn_hip.so r5=tools/ab_libs/libqattn_r5.so (round 5's product library; a missing library is an error)
  --token: token-wise scales of q and k
  name=path@VAR=VAL[@VAR2=VAL2]: environment set around that variant's calls (dev library switches that are read per call)
Prints median / min ms per variant and the ratio to the first library's variant of the same (precision, path).
"""
import argparse
import ctypes
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vp, i, f, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
PREC = {"auto": 0, "fast": 1, "accurate": 2}
SCALE_MODE = 0   # --token: 1 (per-row scales of q and k)
FMT = {"e4m3": 0, "e5m2": 1}


def load(path):
    L = ctypes.CDLL(os.path.abspath(path))
    L.qattn_fp8_tensor_bytes.restype = sz; L.qattn_fp8_tensor_bytes.argtypes = [i, i, i, i, i]
    L.qattn_quant_qkv_workspace_bytes.restype = sz; L.qattn_quant_qkv_workspace_bytes.argtypes = [i, i, i]
    L.qattn_quant_qkv_fp8.restype = i
    L.qattn_quant_qkv_fp8.argtypes = [vp, vp, vp, i, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, vp, sz, vp]
    L.qattn_attention_workspace_bytes.restype = sz; L.qattn_attention_workspace_bytes.argtypes = [i, i, i]
    L.qattn_fp8_attention_forward.restype = i
    L.qattn_fp8_attention_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, i, i, f, i, i, vp, sz, vp]
    L.qattn_fp8_quant_attention_workspace_bytes.restype = sz; L.qattn_fp8_quant_attention_workspace_bytes.argtypes = [i, i, i, i]
    L.qattn_fp8_quant_attention_forward.restype = i
    L.qattn_fp8_quant_attention_forward.argtypes = [vp, vp, vp, i, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, i, f, i, vp, sz, vp]
    return L


class Variant:
    def __init__(self, name, L, q, k, v, causal, fp8):
        self.name, self.L, self.q, self.k, self.v, self.causal, self.fp8 = name, L, q, k, v, int(causal), fp8
        B, H, S, D = q.shape
        self.dims = (B, H, S, D)
        dev = q.device
        u8 = lambda n: torch.empty((max(int(n), 16),), dtype=torch.uint8, device=dev)
        self.q8 = u8(B * H * S * D)
        self.kf = u8(L.qattn_fp8_tensor_bytes(1, B, H, S, D)); self.vf = u8(L.qattn_fp8_tensor_bytes(2, B, H, S, D))
        self.sq, self.sk, self.sv = (torch.empty((B, H, S) if (SCALE_MODE and n < 2) else (B, H), dtype=torch.float32, device=dev) for n in range(3))
        self.out = torch.empty_like(q)
        self.ws_q = u8(L.qattn_quant_qkv_workspace_bytes(B, H, H)); self.ws_a = u8(L.qattn_attention_workspace_bytes(B, H, S))
        self.ws_f = u8(L.qattn_fp8_quant_attention_workspace_bytes(B, H, H, S))
        self.st = torch.cuda.current_stream().cuda_stream
        self.quant()
        # a second set of pre-quantised operands for `attn` (the fused call rewrites kf / vf with a block-scaled V)
        self.kf2, self.vf2, self.q82 = self.kf.clone(), self.vf.clone(), self.q8.clone()
        self.sq2, self.sk2, self.sv2 = self.sq.clone(), self.sk.clone(), self.sv.clone()

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{self.name}: {what} -> {rc}")

    def quant(self):
        B, H, S, D = self.dims
        self._chk(self.L.qattn_quant_qkv_fp8(self.q.data_ptr(), self.k.data_ptr(), self.v.data_ptr(), 2, self.q8.data_ptr(), self.kf.data_ptr(),
                                             self.vf.data_ptr(), self.sq.data_ptr(), self.sk.data_ptr(), self.sv.data_ptr(), B, H, H, S, S, D,
                                             self.fp8, SCALE_MODE, 0, self.ws_q.data_ptr(), self.ws_q.numel(), self.st), "quant_qkv")

    def attn(self, prec):
        B, H, S, D = self.dims
        self._chk(self.L.qattn_fp8_attention_forward(self.q82.data_ptr(), self.kf2.data_ptr(), self.vf2.data_ptr(), self.out.data_ptr(), None,
                                                     self.sq2.data_ptr(), self.sk2.data_ptr(), self.sv2.data_ptr(), B, H, H, S, S, D, self.fp8, self.fp8, 2,
                                                     SCALE_MODE, self.causal, 0.0, prec, 0, self.ws_a.data_ptr(), self.ws_a.numel(), self.st), "attention")

    def fused(self, prec):
        B, H, S, D = self.dims
        self._chk(self.L.qattn_fp8_quant_attention_forward(self.q.data_ptr(), self.k.data_ptr(), self.v.data_ptr(), 2, self.out.data_ptr(),
                                                           self.q8.data_ptr(), self.kf.data_ptr(), self.vf.data_ptr(), self.sq.data_ptr(), self.sk.data_ptr(),
                                                           self.sv.data_ptr(), B, H, H, S, S, D, self.fp8, SCALE_MODE, 0, self.causal, 0.0, prec,
                                                           self.ws_f.data_ptr(), self.ws_f.numel(), self.st), "fused step")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--shape", default="4,32,4096,128")
    ap.add_argument("--causal", action="store_true")
    ap.add_argument("--prec", default="auto,fast")
    ap.add_argument("--paths", default="fused,attn")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--calls", type=int, default=20)
    ap.add_argument("--fp8", default="e4m3")
    ap.add_argument("--scale", type=float, default=1.0, help="multiplies q (score spread)")
    ap.add_argument("--settle", type=float, default=0.5)
    ap.add_argument("--token", action="store_true", help="token-wise scales of q and k (QATTN_SCALE_TOKEN)")
    a = ap.parse_args()
    global SCALE_MODE
    SCALE_MODE = 1 if a.token else 0
    libs, envs = [], {}
    for x in a.libs:
        n, rest = x.split("=", 1)
        parts = rest.split("@")
        libs.append([n, parts[0]])
        envs[n] = dict(e.split("=", 1) for e in parts[1:])
    if not libs:   # default: the product library against the previous round's (tools/ab_libs travels to the GPU box: it is NOT in .gpurunignore)
        libs = [["new", os.path.join(ROOT, "quantumattention_amd", "libqattn_hip.so")], ["r5", os.path.join(ROOT, "tools", "ab_libs", "libqattn_r5.so")]]
    for n, p in libs:   # never silently drop a column: a missing baseline is an error
        if not os.path.exists(p):
            sys.exit(f"tools/ab.py: library {n}={p} does not exist (build it, e.g. from the previous round's tag: see tools/README.md)")
    B, H, S, D = (int(x) for x in a.shape.split(","))
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    if a.scale != 1.0:
        q = (q.float() * a.scale).to(torch.bfloat16)
    loaded = {}
    vs = [Variant(n, loaded.setdefault(p, load(p)), q, k, v, a.causal, FMT[a.fp8]) for n, p in libs]
    jobs = []
    for path in a.paths.split(","):
        for prec in (a.prec.split(",") if path != "quant" else ["-"]):
            for x in vs:
                raw = x.quant if path == "quant" else (lambda x=x, p=PREC[prec], path=path: getattr(x, path)(p))

                def fn(raw=raw, env=envs.get(x.name, {})):
                    for k_, v_ in env.items():
                        os.environ[k_] = v_
                    raw()
                    for k_ in env:
                        del os.environ[k_]
                jobs.append((f"{path:5s} {prec:8s} {x.name}", (path, prec), fn))
    t_end = time.time() + a.settle          # settle the clocks under load (the chip idles at 102 MHz)
    while time.time() < t_end:
        for _, _, fn in jobs:
            fn()
        torch.cuda.synchronize()
    times = {n: [] for n, _, _ in jobs}
    for r in range(a.rounds):
        order = jobs if r % 2 == 0 else jobs[::-1]
        for n, _, fn in order:
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.calls):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[n].append(e0.elapsed_time(e1) / a.calls)
    base = {}
    print(f"shape B{B} H{H} S{S} D{D} causal={a.causal} fp8={a.fp8} q x{a.scale}  ({a.rounds} rounds x {a.calls} calls, interleaved)")
    for n, key, _ in jobs:
        med, mn = statistics.median(times[n]), min(times[n])
        base.setdefault(key, med)
        print(f"  {n:28s} median {med:.4f} ms  min {mn:.4f}  ({med / base[key]:.3f} x {libs[0][0]})")


if __name__ == "__main__":
    main()
