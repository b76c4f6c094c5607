#!/usr/bin/env python3
"""Headline benchmark: FP8 fused attention forward on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: `fp8_attn_func(q, k, v)` on bf16 [B,H,S,D] inputs already
resident in HBM, i.e. the bf16->fp8 quant pre-pass of q, k, v AND the fused attention kernel (what the
reference's own benchmark times, tests/test_interface.py:104-108,136).  Workload = BASELINE.json configs[1]:
B=4 H=32 S=4096 D=128 non-causal e4m3 per GPU; with --gpus N every rank runs that shard (batch-sharded
B = 4N, configs[3] at N=8; no collective on the data path -- SURVEY.md §8e), so scaling is "weak".

FLOPs follow the reference convention 4*B*H*Sq*Skv*D (tests/test_interface.py:121-125).
The JSON line also carries
  roofline     -- the dominant kernel (attn_fwd_kernel) alone: algorithmic FLOPs per launch / its average launch
                  duration measured with HIP events on the launch stream, against the 5.0 PFLOP/s dense fp8 MFMA peak;
  cpu_baseline -- the reference's CPU path (torch port of ops.py:64-95, oracle/torch_ref.py) timed on the host cores
                  on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP8_PEAK_TFLOPS = 5000.0  # MI355X dense fp8 MFMA peak (MI355X_MICROARCH.md: ~5 PF dense)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--seq", type=int, default=4096)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--causal", action="store_true")
    ap.add_argument("--fp8", default="e4m3", choices=["e4m3", "e5m2"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def flops(B, H, Sq, Skv, D, causal):
    f = 4.0 * B * H * Sq * Skv * D
    return f / 2 if causal else f


def cpu_baseline(args, q, k, v):
    """Reference CPU path on a bounded sample: one batch element (H heads) of the same workload."""
    from oracle import torch_ref

    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    qs, ks, vs = q[:1].cpu(), k[:1].cpu(), v[:1].cpu()
    q8, sq = torch_ref.quantize_fp8_eager_ref(qs, reduction_dim=[2, 3])
    k8, sk = torch_ref.quantize_fp8_eager_ref(ks, reduction_dim=[2, 3])
    torch_ref.fp8_attention_forward_ref(q8[:, :2], k8[:, :2], vs[:, :2], sq[:, :2], sk[:, :2], is_causal=args.causal)
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        torch_ref.fp8_attention_forward_ref(q8, k8, vs, sq, sk, is_causal=args.causal)
        best = min(best, time.perf_counter() - t0)
    f = flops(1, args.heads, args.seq, args.seq, args.dim, args.causal)
    return {
        "value": f / best / 1e12, "unit": "TFLOP/s", "cores": threads, "kind": "port",
        "sample": f"B=1 H={args.heads} S={args.seq} D={args.dim} (1/{args.batch} of one GPU's batch), best of 3, "
                  f"torch {torch.__version__} CPU bf16 SDPA on de-quantised q,k (ops.py:64-95)",
        "seconds": best,
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    os.environ.setdefault("QATTN_STEP_EVENTS", "1")  # the library brackets attention launches with HIP events (bench only)
    import quantumattention_amd as qa
    from quantumattention_amd import _native

    B, H, S, D = args.batch, args.heads, args.seq, args.dim
    torch.manual_seed(rank)
    q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))

    def step():
        with qa.config.patch({"attention.fp8_format": args.fp8}):
            return qa.fp8_attn_func(q, k, v, is_causal=args.causal)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- dominant kernel alone (roofline): pre-quantised operands, HIP events on the launch stream
    fp8_dtype = _native.FP8_DTYPE[args.fp8]
    q8, kf, vf, sq, sk, sv = _native.quant_qkv_fp8(q, k, v, fp8_dtype=fp8_dtype)

    def attn_only():
        return _native.fp8_attention_forward(q8, kf, vf, sq, sk, sv, Hkv=H, Skv=S, out_dtype=torch.bfloat16,
                                             is_causal=args.causal)

    def quant_only():
        _native.quant_qkv_fp8(q, k, v, fp8_dtype=fp8_dtype)

    def event_time(fn, n):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n  # ms per launch

    def attn_in_step(n):
        """The attention launch's duration INSIDE the real step (the fused C entry: pre-pass -> attention), from the two HIP
        events the library records on the launch stream around the attention launch (QATTN_STEP_EVENTS=1)."""
        for _ in range(3):
            step()
        tot = 0.0
        for _ in range(n):
            step()
            ms = _native.lib().qattn_debug_last_attention_ms()   # waits for this step's attention to finish
            if ms < 0:
                return None
            tot += ms
        return tot / n

    attn_isolated_ms = event_time(attn_only, args.steps)   # the kernel launched back to back on pre-quantised operands
    attn_ms = attn_in_step(args.steps)                     # what the roofline is computed from: the launch inside the step
    if attn_ms is None:
        attn_ms = attn_isolated_ms
    quant_ms = event_time(quant_only, args.steps)
    # informational: the same step replayed from a HIP graph (no launch gaps); `value` stays the eager API call
    graph_ms = None
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        graph_ms = event_time(g.replay, args.steps)
    except Exception as exc:  # capture is optional evidence, never fatal for the benchmark line
        print(f"[bench] HIP graph capture skipped: {exc}", file=sys.stderr)

    if rank == 0:
        f_gpu = flops(B, H, S, S, D, args.causal)
        value = f_gpu * world * args.steps / elapsed / 1e12
        achieved = f_gpu / (attn_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and not args.causal and (B, H, S, D) == (4, 32, 4096, 128):
            traffic = json.load(open(tpath)).get("attn_fwd_hbm_bytes_per_launch")
        line = {
            "metric": "attention fwd TFLOP/s (fp8), quant pre-pass + fused attention",
            "value": value, "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": f"fp8_{args.fp8} (fp32 accumulate)", "data": "synthetic",
            "config": {"workload": f"B={B} H={H} S={S} D={D} {'causal' if args.causal else 'non-causal'} fp8({args.fp8}) "
                                   f"per GPU, bf16 in/out, head-wise scales (BASELINE.json configs[1]"
                                   f"{'; batch-sharded B=%d total' % (B * world) if world > 1 else ''})",
                       "global_batch": B * world, "parallelism": f"batch-shard x{world}, no collectives"},
            "frac_of_fp8_mfma_peak": value / (FP8_PEAK_TFLOPS * world),
            "attn_kernel_ms": attn_ms, "attn_kernel_isolated_ms": attn_isolated_ms, "quant_prepass_ms": quant_ms, "graph_replay_ms_per_step": graph_ms,
            "roofline": {"kernel": "qattn::attn_fwd_kernel_v2<128,8,e4m3,e4m3,...,Q16> (fused QK^T/softmax/PV; timed inside the step)", "bound": "mfma", "achieved": achieved,
                         "peak": FP8_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP8_PEAK_TFLOPS,
                         "traffic": traffic},
        }
        if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N=1 only
            line["cpu_baseline"] = cpu_baseline(args, q, k, v)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
