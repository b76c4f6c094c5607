"""The N>1 path on CPU: the product's own sharding helpers (quantumattention_amd/utils/shard.py) under
two gloo ranks, and bench.py's own multi-rank launcher in --dry-run mode.

(1) every rank draws ITS shard with the helper bench.py uses; rank 0 also draws the unsharded batch: the shards must be
    the exact slices, and attending them separately must equal the slices of the unsharded result -- computed here with
    the CPU oracle since there is no GPU (on the GPU the same equivalence is checked bit-exactly on the product kernels,
    tests/test_gpu_attention.py::test_full_size_properties_*).  Also the timing reduction (barrier + MAX all-reduce).
(2) `python bench.py --gpus 2 --dry-run` must start two ranks by itself and print ONE line with n_gpus == ranks_seen == 2."""
import json
import os
import subprocess
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
    from quantumattention_amd.utils.shard import batch_shard, synthetic_qkv

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B_total, H, S, D = 4, 2, 96, 64
    shard = batch_shard(B_total, rank, world)
    q, k, v = synthetic_qkv(shard, H, S, D, seed=7)           # what bench.py does on every rank
    b16 = lambda t: t.contiguous().view(torch.int16).numpy().view(np.uint16)

    def attend(q, k, v):
        (q8, sq), (k8, sk), (v8, sv) = (oracle.quantize_fp8(b16(x), oracle.FMT_BF16, "head") for x in (q, k, v))
        return oracle.attention_forward(q8, k8, v8, 0, 0, 0, sq, sk, sv, causal=True)

    np.save(os.path.join(out_dir, f"o_{rank}.npy"), attend(q, k, v))
    np.save(os.path.join(out_dir, f"q_{rank}.npy"), b16(q))
    # bench.py's timing reduction: barrier, then MAX over ranks of the elapsed time, and the rank census
    dist.barrier()
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == 0.5 + (world - 1)
    ones = torch.ones(1, dtype=torch.float64)
    dist.all_reduce(ones)
    assert int(ones) == world
    if rank == 0:
        qf, kf, vf = synthetic_qkv(batch_shard(B_total, 0, 1), H, S, D, seed=7)   # the unsharded batch
        np.save(os.path.join(out_dir, "o_full.npy"), attend(qf, kf, vf))
        np.save(os.path.join(out_dir, "q_full.npy"), b16(qf))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_batch_shard_equals_unsharded(tmp_path):
    world, port = 2, 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    full, qfull = np.load(tmp_path / "o_full.npy"), np.load(tmp_path / "q_full.npy")
    per = full.shape[0] // world
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"q_{r}.npy"), qfull[r * per:(r + 1) * per])
        np.testing.assert_array_equal(np.load(tmp_path / f"o_{r}.npy"), full[r * per:(r + 1) * per])


def test_shard_helper_rejects_uneven_splits():
    from quantumattention_amd.utils.shard import batch_shard

    assert list(batch_shard(32, 3, 8)) == [12, 13, 14, 15]     # BASELINE config 4: B=32 over 8 GPUs, 4 per GPU
    with pytest.raises(ValueError):
        batch_shard(6, 0, 4)
    with pytest.raises(ValueError):
        batch_shard(8, 8, 8)


@pytest.mark.parametrize("gpus", [1, 2])
def test_bench_launches_its_own_ranks_dry_run(gpus):
    """`python bench.py --gpus N` needs no wrapper.  --dry-run: gloo + a host stub instead of the device step."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--dry-run", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == gpus and d["ranks_seen"] == gpus and d["steps"] == 3 and d["warmup"] == 1
    assert d["config"]["global_batch"] == 4 * gpus and d["scaling"] == "weak" and d["value"] > 0


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29431")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert p.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in (p.stderr + p.stdout)


def test_bench_under_torch_distributed_run_dry_run():
    """The driver's N > 1 command line: torch.distributed.run starts the ranks, bench.py takes RANK / WORLD_SIZE from the environment."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    port = 29700 + (os.getpid() % 200)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["config"]["global_batch"] == 8
