#!/usr/bin/env python3
"""Randomised launch-structure sweep (no oracle needed): shapes large enough for the persistent / dynamic launches of the D = 128
kernel and for the multi-launch paths of the templated one -- B x Hq not a multiple of 8 (no XCD map), GQA, odd block counts,
ragged lengths, causal and not, every precision, head- and token-wise -- each checked four ways:
  1. every output element is written (the output buffer is poisoned with NaN first, through the allocator);
  2. the batched call equals the calls on single batch elements bit for bit (those take the small-launch paths);
  3. a HIP graph of the call, replayed twice on new data, equals the eager call bit for bit;
  4. (head-wise) the call with the caller-supplied exact per-head abs-max of q and k equals the plain call bit for bit.
  python tools/fuzz_launch.py [N=40] [seed=0]"""
import os, sys, time
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantumattention_amd as qa  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
fails = 0
t0 = time.time()
for case in range(N):
    D = int(rng.choice([128, 128, 128, 64, 256]))
    causal = bool(rng.integers(2))
    Hkv = int(rng.choice([1, 2, 3, 4, 5, 8]))
    Hq = Hkv * int(rng.choice([1, 1, 2, 4]))
    B = int(rng.choice([1, 2, 3, 4, 6, 8, 12]))
    S = int(rng.choice([rng.integers(1024, 2200), rng.integers(2200, 4200), rng.integers(4200, 9000)]))
    while B * Hq * S * D > 5.0e8 or B * Hq * S * S > 6.0e10:
        S = S * 3 // 4
    # enough query blocks for the persistent launch (> 256 workgroups of 256 rows) in most cases
    blocks = B * Hq * ((S + 255) // 256)
    scaling = str(rng.choice(["head-wise", "head-wise", "head-wise", "token-wise"]))
    precision = str(rng.choice(["auto", "auto", "fast", "accurate"]))
    fp8 = str(rng.choice(["e4m3", "e4m3", "e5m2"]))
    dtype = torch.bfloat16 if rng.integers(4) else torch.float16
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    q = (torch.randn(B, Hq, S, D, generator=g, device="cuda") * float(rng.choice([1.0, 1.0, 1.3, 2.0]))).to(dtype)
    k = torch.randn(B, Hkv, S, D, generator=g, device="cuda").to(dtype)
    v = torch.randn(B, Hkv, S, D, generator=g, device="cuda").to(dtype)
    fn = qa.fp8_attn_func if scaling == "head-wise" else qa.fp8_token_wise_attn_func
    msg = []
    with qa.config.patch({"attention.precision": precision, "attention.fp8_format": fp8}):
        junk = torch.full_like(q, float("nan")); del junk
        whole = fn(q, k, v, is_causal=causal)
        if not torch.isfinite(whole).all():
            msg.append("unwritten / non-finite output")
        for b in range(B):
            if not torch.equal(whole[b:b + 1], fn(q[b:b + 1], k[b:b + 1], v[b:b + 1], is_causal=causal)):
                msg.append(f"batch element {b} differs from its own call")
                break
        if case % 3 == 0:   # the same values as transposed views of [B,S,H,D] tensors, output in the query's layout (include/qattn_strided.h):
            #                     the persistent / dynamic launches of the strided-view instantiations
            qv, kv, vv = (t.transpose(1, 2).contiguous().transpose(1, 2) for t in (q, k, v))
            with qa.config.patch({"attention.output_layout": "like_query" if case % 2 else "contiguous"}):
                if not torch.equal(whole, fn(qv, kv, vv, is_causal=causal)):
                    msg.append("call on strided views differs from the dense call")
            del qv, kv, vv
        if scaling == "head-wise" and case % 2 == 1:   # producer hand-off: the exact per-head abs-max supplied by the caller
            # (under AUTO the sums of squares as well: without them the kernel has no score-spread estimate and wide heads start
            #  one-term -- the same bound, other bits; the fp32 sums differ from the pass's partial sums in the last bits, which
            #  only matters for a head exactly on the dead band's edge)
            aq, ak = q.abs().amax(dim=(2, 3)).float(), k.abs().amax(dim=(2, 3)).float()
            extra = dict(ssq_q=(q.float() ** 2).sum(dim=(2, 3)), ssq_k=(k.float() ** 2).sum(dim=(2, 3))) if precision == "auto" else {}
            got = fn(q, k, v, is_causal=causal, amax_q=aq, amax_k=ak, **extra)   # (the keyword form of the reference-shaped interface)
            if not torch.equal(whole, got):
                msg.append("call with producer-supplied abs-max (and sums of squares) differs")
        if case % 3 == 0:   # graph capture + two replays on new data
            side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn(q, k, v, is_causal=causal)
            torch.cuda.current_stream().wait_stream(side)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                og = fn(q, k, v, is_causal=causal)
            for r in range(2):
                q.copy_((torch.randn(q.shape, generator=g, device="cuda")).to(dtype)); k.copy_(torch.randn(k.shape, generator=g, device="cuda").to(dtype))
                gr.replay(); torch.cuda.synchronize()
                if not torch.equal(og, fn(q, k, v, is_causal=causal)):
                    msg.append(f"graph replay {r} differs from the eager call")
                    break
            del gr, og
    fails += bool(msg)
    print(f"{'ok  ' if not msg else 'FAIL'} #{case:3d} B{B} Hq{Hq} Hkv{Hkv} S{S} D{D} {'causal' if causal else 'full  '} {fp8} {scaling[:5]} "
          f"{'bf16' if dtype == torch.bfloat16 else 'fp16'} {precision:8s} blocks {blocks:5d}{' +graph' if case % 3 == 0 else ''} {'; '.join(msg)}", flush=True)
    del q, k, v, whole
print(f"{N} cases, {fails} failures, {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
