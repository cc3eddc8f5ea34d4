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


# ---- the real sharded entry point (config 4) through gloo, with a stub in place of the GPU sampler --------------------
class _StubTrainer:
    """DDPMTrainer with generate_batch replaced by a CPU function of (conditioning window, chain key, window seed, gt
    hand-off): everything above generate_batch — get_windows, the chain loop, split_segments, shard_range, batching of
    equal-length chains, gather_outputs — is the product code."""

    def __new__(cls, cfg):
        from diffsheg_amd.trainer import DDPMTrainer, sampler_namespace

        class Stub(DDPMTrainer):
            def __init__(self, opt):                       # no native handle: the sampler is stubbed
                self.opt, self.device = opt, torch.device("cpu")

            def generate_batch(self, audio_emb, p_id, dim_pose, add_cond={}, inpaint_dict=None, seed=None, row_keys=None, **kw):
                B, T = audio_emb.shape[:2]
                out = torch.empty(B, T, dim_pose)
                for b in range(B):
                    g = torch.Generator().manual_seed((int(seed) * 1000003 + int(row_keys[b])) & ((1 << 62) - 1))
                    out[b] = (torch.randn(T, dim_pose, generator=g) + audio_emb[b].mean(-1, keepdim=True)
                              + add_cond["pretrain_aud_feat"][b].mean(-1, keepdim=True) + p_id[b].argmax())
                m = (inpaint_dict or {}).get("outpainting_mask")
                if m is not None and bool(m.any()):
                    out = torch.where(m, 0.5 * inpaint_dict["gt"] + 0.5 * out, out)
                return out
        return Stub(sampler_namespace(cfg))


def _stream_inputs(cfg, N):
    g = torch.Generator().manual_seed(5)
    return (torch.randn(1, N, cfg.audio_dim, generator=g), {"pretrain_aud_feat": torch.randn(1, N, 16, generator=g)},
            torch.eye(cfg.style_dim)[1:2])


def _sharded_worker(rank, world, port, N, n_seg, rank0_only, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from diffsheg_amd.config import get_config
        cfg = get_config("show")
        tr = _StubTrainer(cfg)
        audio, cond, pid = _stream_inputs(cfg, N)
        if rank0_only and rank != 0:
            audio, cond = None, None
        out = tr.sample_arbitrary_len_sharded(audio, pid, cond, n_seg, seed=11, inputs_on_rank0_only=rank0_only)
        assert (out is None) == (rank != 0)
        # by VALUE (a numpy array is pickled into the queue's pipe; a torch tensor would travel as a shared-memory handle that
        # dies with this process: the parent then fails to rebuild it when the worker has already exited)
        q.put((rank, None if out is None else out.numpy().copy()))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("N,n_seg,rank0_only", [(9000, 8, False), (1000, 5, True), (300, 3, False)])
def test_sharded_long_audio_world2_equals_single_rank_per_chain(N, n_seg, rank0_only):
    from diffsheg_amd.config import get_config
    from diffsheg_amd.trainer import split_segments
    cfg = get_config("show")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, N, n_seg, rank0_only, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out2 = torch.from_numpy(got[0])
    assert out2.shape == (1, N, cfg.net_dim_pose)
    # single rank, same entry point
    tr = _StubTrainer(cfg)
    audio, cond, pid = _stream_inputs(cfg, N)
    out1 = tr.sample_arbitrary_len_sharded(audio, pid, cond, n_seg, seed=11)
    assert torch.equal(out1, out2)
    # every chain on its own (batch of one) equals its slice of the gathered stream: batching / sharding changed nothing
    segs = split_segments(N, n_seg, cfg.n_poses, cfg.overlap_len)
    assert sum(len(s) for s in segs) == N and segs[0].start == 0 and segs[-1].stop == N
    for i, sg in enumerate(segs):
        solo = tr.sample_arbitrary_len(audio[:, sg.start:sg.stop], pid, {k: v[:, sg.start:sg.stop] for k, v in cond.items()},
                                       seed=11, row_keys=[i])
        assert torch.equal(solo[0], out2[0, sg.start:sg.stop]), i


@pytest.mark.parametrize("mode", ["batch", "chain", "ddpm"])
def test_bench_gpus_n_launches_n_ranks_itself(mode):
    """`python bench.py --gpus 2` started WITHOUT a launcher (how the driver starts N = 1) must run two ranks and report
    n_gpus = 2 — never a silent single-rank line.  DSH_BENCH_DRYRUN=1 stops each rank after the rendezvous + barrier +
    max-reduce of the timing contract (there is no GPU here); the real thing runs in tests/test_gpu_sharded.py."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DSH_BENCH_DRYRUN="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--mode", mode, "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-300:], r.stderr[-800:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["max_over_ranks"] == 2.0 and d["mode"] == mode
    # a launcher that provides a different world size than --gpus is refused, not silently accepted
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--mode", mode], env=env2, capture_output=True, text=True,
                        timeout=300)
    assert r2.returncode != 0 and not [ln for ln in r2.stdout.splitlines() if ln.startswith("{")]
