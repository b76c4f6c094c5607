"""The N>1 path of bench.py on CPU: two gloo ranks, batch-sharded, no data-path collective.

Each rank owns a batch shard; the test checks (a) the shard decomposition is exact -- every rank's output equals the
corresponding slice of the unsharded result, computed here with the CPU oracle since there is no GPU -- and (b) the
timing reduction bench.py uses (barrier + MAX all-reduce of the per-rank elapsed time) works under gloo."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import oracle

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B_total, H, S, D = 4, 2, 96, 64
    per = B_total // world
    torch.manual_seed(0)  # every rank generates the full batch, then keeps its shard (bench.py seeds per rank instead)
    q, k, v = (torch.randn(B_total, H, S, D, dtype=torch.bfloat16) for _ in range(3))
    sl = slice(rank * per, (rank + 1) * per)
    b16 = lambda t: t.contiguous().view(torch.int16).numpy().view(np.uint16)
    outs = []
    for x in (q[sl], k[sl], v[sl]):
        outs.append(oracle.quantize_fp8(b16(x), oracle.FMT_BF16, "head"))
    (q8, sq), (k8, sk), (v8, sv) = outs
    o = oracle.attention_forward(q8, k8, v8, 0, 0, 0, sq, sk, sv, causal=True)
    np.save(os.path.join(out_dir, f"o_{rank}.npy"), o)
    # bench.py's timing reduction: barrier, then MAX over ranks of the elapsed time
    dist.barrier()
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == 0.5 + (world - 1)
    if rank == 0:
        # unsharded reference on rank 0
        full = []
        for x in (q, k, v):
            full.append(oracle.quantize_fp8(b16(x), oracle.FMT_BF16, "head"))
        (Q8, SQ), (K8, SK), (V8, SV) = full
        np.save(os.path.join(out_dir, "o_full.npy"), oracle.attention_forward(Q8, K8, V8, 0, 0, 0, SQ, SK, SV, causal=True))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_batch_shard_equals_unsharded(tmp_path):
    world, port = 2, 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    full = np.load(tmp_path / "o_full.npy")
    per = full.shape[0] // world
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"o_{r}.npy"), full[r * per:(r + 1) * per])
