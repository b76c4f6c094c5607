"""Batch sharding of the hot path over the GPUs of one node.

Every (batch, head) of the attention call is independent (the reference's grid is (q-block, head, batch) with no
cross-CTA traffic, src/quantum_attn/tk/attention.py:504), so the multi-GPU decomposition is: split dim 0 (batch) evenly,
one process per GPU, no collective on the data path.  bench.py and the multi-process tests share these helpers."""
from typing import Tuple

import torch


def batch_shard(global_batch: int, rank: int, world: int) -> range:
    """Batch indices owned by `rank`: contiguous, equal sizes (global_batch must divide evenly, as BASELINE config 4 does)."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world: {rank}/{world}")
    if global_batch % world != 0:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return range(rank * per, (rank + 1) * per)


def synthetic_qkv(shard: range, heads: int, seq: int, dim: int, *, dtype=torch.bfloat16, device="cpu", seed: int = 0,
                  kv_heads: int = 0, kv_seq: int = 0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """N(0,1) q, k, v for the batch elements of `shard` (the reference's test distribution, tests/test_interface.py:40-43).
    Element b of the GLOBAL batch is drawn from its own generator seeded with (seed, b), so a rank's shard is exactly the
    corresponding slice of the unsharded tensors whatever the world size."""
    kv_heads = kv_heads or heads
    kv_seq = kv_seq or seq
    qs, ks, vs = [], [], []
    for b in shard:
        g = torch.Generator(device="cpu")
        g.manual_seed(seed * 1_000_003 + b)
        qs.append(torch.randn(heads, seq, dim, generator=g, dtype=torch.float32).to(dtype))
        ks.append(torch.randn(kv_heads, kv_seq, dim, generator=g, dtype=torch.float32).to(dtype))
        vs.append(torch.randn(kv_heads, kv_seq, dim, generator=g, dtype=torch.float32).to(dtype))
    return tuple(torch.stack(t).to(device) for t in (qs, ks, vs))
