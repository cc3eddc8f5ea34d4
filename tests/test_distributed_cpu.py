"""world_size-2 gloo tests (CPU) of the multi-rank plumbing: contiguous sharding of independent
clips / chains, padded gather to rank 0, and the bench contract's barrier + max-reduce timing."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffsheg_amd.trainer import gather_outputs, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard_range(n_items, rank, world)
        # each "clip" i produces a [T=3, C=2] block filled with i (stands in for a sampled chain)
        local = torch.stack([torch.full((3, 2), float(i)) for i in mine]) if len(mine) else torch.zeros(0, 3, 2)
        sizes = [len(shard_range(n_items, r, world)) for r in range(world)]
        parts = gather_outputs(local, sizes)
        # timing contract: max over ranks of a per-rank scalar
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            full = torch.cat(parts)
            q.put((full[:, 0, 0].tolist(), float(t)))
        else:
            assert parts is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [7, 2, 1])
def test_shard_and_gather_world2(n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    vals, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert vals == [float(i) for i in range(n_items)]
    assert tmax == 2.0
