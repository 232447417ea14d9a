"""The N>1 path on CPU: world_size-2 gloo processes each take a batch shard (dense and ragged), nothing is exchanged
on the data path, and the union of the shards equals the unsharded result (computed with the oracle here — the
decomposition is what is under test; the HIP kernel itself is covered by the -m gpu tests).  Also covers the
max-over-ranks timing reduction bench.py uses."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from flash_attention_annotated_amd import sharding
from oracle import attention_ref as oracle


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)  # every rank builds the same global problem, then keeps only its shard
        q = torch.randn(5, 33, 4, 16)
        k = torch.randn(5, 47, 2, 16)
        v = torch.randn(5, 47, 2, 16)
        qs, ks, vs = sharding.shard_dense(q, k, v, rank, world)
        out_s, _ = oracle.attention_ref(qs, ks, vs, causal=True)
        # ragged batch
        lens = [9, 0, 31, 17, 5]
        cu = torch.tensor([0, 9, 9, 40, 57, 62], dtype=torch.int32)
        qv = torch.randn(62, 4, 16)
        kv = torch.randn(62, 2, 16)
        vv = torch.randn(62, 2, 16)
        q2, k2, v2, cq, ck, mq, mk = sharding.shard_varlen(qv, kv, vv, cu, cu, rank, world)
        assert int(cq[0]) == 0 and mq == (max(lens[slice(*sharding.shard_range(5, rank, world))]) if q2.shape[0] else 0)
        outv_s, _ = oracle.attention_varlen_ref(q2, k2, v2, cq, ck)
        torch.save({"dense": out_s, "varlen": outv_s}, os.path.join(tmpdir, f"rank{rank}.pt"))
        # timing reduction: max over ranks (the only collective, and it is not on the data path)
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == float(world)
        dist.barrier()
        if rank == 0:
            full, _ = oracle.attention_ref(q, k, v, causal=True)
            fullv, _ = oracle.attention_varlen_ref(qv, kv, vv, cu, cu)
            parts = [torch.load(os.path.join(tmpdir, f"rank{r}.pt"), weights_only=True) for r in range(world)]
            assert torch.equal(torch.cat([p["dense"] for p in parts]), full)
            assert torch.equal(torch.cat([p["varlen"] for p in parts]), fullv)
            assert sharding.aggregate_throughput(10.0, world, 2.0) == 10.0 * world / 2.0
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_batch_shard_two_ranks_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
