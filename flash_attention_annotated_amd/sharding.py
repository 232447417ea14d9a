"""Multi-GPU decomposition of the forward: plain batch sharding, no collective.

Every (batch, head, m_block) tile of the forward is independent (hopper/tile_scheduler.hpp:84-86: no cross-tile
reduction without split-KV), so N GPUs each take a contiguous chunk of the batch (for ragged batches: a contiguous
range of sequences with `cu_seqlens` re-based to 0) and nothing is exchanged.  SURVEY.md §8(e).
"""
from typing import Tuple

import torch


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, stop) of `n` items owned by `rank`: contiguous, sizes differ by at most one, union = range(n)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_dense(q, k, v, rank: int, world: int):
    """Batch shard of dense (b, s, h, d) tensors (views, no copy)."""
    a, b = shard_range(q.shape[0], rank, world)
    return q[a:b], k[a:b], v[a:b]


def shard_varlen(q, k, v, cu_seqlens_q, cu_seqlens_k, rank: int, world: int):
    """Sequence-range shard of a packed ragged batch.  Returns (q, k, v, cu_q, cu_k, max_q, max_k) with the
    cumulative lengths re-based to 0 (int32, same device)."""
    nseq = cu_seqlens_q.numel() - 1
    a, b = shard_range(nseq, rank, world)
    cq = cu_seqlens_q[a:b + 1]
    ck = cu_seqlens_k[a:b + 1]
    q0, q1 = int(cq[0]), int(cq[-1])
    k0, k1 = int(ck[0]), int(ck[-1])
    cq = (cq - cq[0]).to(torch.int32)
    ck = (ck - ck[0]).to(torch.int32)
    lens_q = cq[1:] - cq[:-1]
    lens_k = ck[1:] - ck[:-1]
    max_q = int(lens_q.max()) if lens_q.numel() else 0
    max_k = int(lens_k.max()) if lens_k.numel() else 0
    return q[q0:q1], k[k0:k1], v[k0:k1], cq, ck, max_q, max_k


def aggregate_throughput(units_per_rank: float, world: int, elapsed_max_s: float) -> float:
    """Whole-job rate: what all ranks processed / the slowest rank's wall time."""
    return units_per_rank * world / elapsed_max_s
