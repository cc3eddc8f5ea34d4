"""Config 4's real code path on a real MI355X: independent chains of one stream batched as batch rows with
per-row Philox streams (dsh_sample_set_row_keys), the sharded entry point, the per-(window, chain) noise keys, the full-batch
ddim25 loops of configs 2 / 3 against their own rows sampled alone, and bench.py's multi-rank launch.

Reference behaviour replaced: one chain per rank from each rank's global torch RNG (ddpm_show_trainer.py:743-750, 864-906,
924-931)."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

from diffsheg_amd import _lib  # noqa: E402
from diffsheg_amd.config import get_config  # noqa: E402
from diffsheg_amd.synthetic import make_inputs  # noqa: E402
from diffsheg_amd.trainer import DDPMTrainer, sampler_namespace, split_segments, window_seed  # noqa: E402
from util import gpu_model, rel_err  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rows(seed, keys, n_row, offset=0):
    out = torch.empty(len(keys), n_row, device="cuda:0")
    karr = (C.c_uint64 * len(keys))(*[int(k) for k in keys])
    _lib.check(_lib.lib().dsh_op_philox_randn_rows(None, out.data_ptr(), len(keys), n_row, seed & (2 ** 64 - 1), offset, karr))
    return out.cpu()


def test_row_streams_do_not_collide_across_windows_and_chains():
    """x_T of (window w, chain s) for w, s < 8 under the harness's key derivation: 64 distinct streams.  Round 2 derived the key
    as (seed + w) ^ s, so that e.g. (w = 1, s = 0) and (w = 0, s = 1) drew identical noise."""
    n_row, seed = 88 * 232, 2024
    xs = torch.stack([_rows(window_seed(seed, w), list(range(8)), n_row) for w in range(8)]).reshape(64, n_row)
    assert torch.isfinite(xs).all()
    assert abs(float(xs.mean())) < 5e-3 and abs(float(xs.std()) - 1.0) < 5e-3
    z = (xs - xs.mean(1, keepdim=True)) / xs.std(1, keepdim=True)
    corr = (z @ z.T) / n_row
    off = corr - torch.diag(torch.diag(corr))
    print(f"[philox rows] max |corr| between the 64 (window, chain) streams: {float(off.abs().max()):.4f}")
    assert float(off.abs().max()) < 0.05
    # the regression itself, at the level of raw (seed, key) pairs: neighbouring seeds with swapped row keys
    a, b = _rows(seed + 1, [0], n_row), _rows(seed, [1], n_row)
    assert not torch.equal(a, b) and abs(float((a * b).mean())) < 0.05
    # a row's stream does not depend on the batch it is drawn in, and later draws (offset) continue it
    solo = _rows(seed, [5], n_row)
    batch = _rows(seed, [3, 5, 9], n_row)
    assert torch.equal(solo[0], batch[1])
    assert not torch.equal(_rows(seed, [5], n_row, offset=n_row // 4)[0], solo[0])


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_sharded_stream_equals_its_chains_sampled_alone(precision):
    """sample_arbitrary_len_sharded (world 1: all chains on this GPU, equal-length ones batched) vs every chain sampled alone
    with row_keys = [its id]: same noise bit for bit, so the chains agree to the round-off a different batch size may cause in
    the small-row GEMMs (fp32 1e-5 of range; bf16: the end-to-end bf16 gate)."""
    cfg = get_config("show")
    model = gpu_model("show", precision)
    tr = DDPMTrainer(sampler_namespace(cfg), model)
    N, n_seg = 5 * 78 + 10 + 37, 5                  # five chains: four of one window + stride each ... and a longer last one
    inp = make_inputs(cfg, 1, frames=N, seed=15)
    audio, cond, pid = inp["audio_emb"], {"pretrain_aud_feat": inp["pretrain_aud_feat"]}, inp["person_id"]
    out = tr.sample_arbitrary_len_sharded(audio, pid, cond, n_seg, seed=31)
    assert out.shape == (1, N, cfg.net_dim_pose) and torch.isfinite(out).all()
    segs = split_segments(N, n_seg, cfg.n_poses, cfg.overlap_len)
    assert len(segs) >= 4
    tol = 1e-5 if precision == "fp32" else 1.2e-2
    worst, exact = 0.0, 0
    for i, sg in enumerate(segs):
        solo = tr.sample_arbitrary_len(audio[:, sg.start:sg.stop].cuda(), pid, {k: v[:, sg.start:sg.stop].cuda() for k, v in cond.items()},
                                       seed=31, row_keys=[i])
        e = rel_err(solo[0], out[0, sg.start:sg.stop])
        worst = max(worst, e)
        exact += int(torch.equal(solo[0], out[0, sg.start:sg.stop]))
    print(f"[sharded {precision}] {len(segs)} chains, {exact} bit-identical to the chain sampled alone, worst rel err {worst:.3e}")
    assert worst < tol
    # different chains are different samples (the conditioning differs AND the noise differs)
    a, b = out[0, segs[0].start:segs[0].start + 60], out[0, segs[1].start:segs[1].start + 60]
    assert not torch.allclose(a, b)


def test_fp32_chain_equality_is_bitwise_without_the_few_row_gemm():
    """Since round 4 a launch of at most 512 rows takes the K-split fp32 GEMM (gemm.hip), whose summation order differs from the tiled
    kernel's: on the fp32 path a chain sampled alone and the same chain inside a larger batch agree to round-off (test above, 5e-7 of
    range), not bit for bit.  DSH_GEMM_KSPLIT=0 is the reproducible mode: with it the equality is bitwise
    again — pinned here in a fresh process (the switch is read once)."""
    env = dict(os.environ, DSH_GEMM_KSPLIT="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fp32_bitwise_worker.py")], env=env, capture_output=True, text=True, timeout=900)
    print(r.stdout[-300:])
    assert r.returncode == 0 and "FP32_BITWISE_OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2000:])


def test_five_minute_stream_32_chains_finishes():
    """BASELINE configs[3]: 9000 frames, overlap 10, 32 independent chains through the sharded entry point, bf16."""
    cfg = get_config("show")
    model = gpu_model("show", "bf16")
    tr = DDPMTrainer(sampler_namespace(cfg), model)
    N = 9000
    inp = make_inputs(cfg, 1, frames=N, seed=3)
    out = tr.sample_arbitrary_len_sharded(inp["audio_emb"].cuda(), inp["person_id"].cuda(), {"pretrain_aud_feat": inp["pretrain_aud_feat"].cuda()},
                                          32, seed=2024)
    assert out.shape == (1, N, cfg.net_dim_pose) and torch.isfinite(out).all()
    segs = split_segments(N, 32, cfg.n_poses, cfg.overlap_len)
    # neighbouring chains start from different noise: their first frames differ by far more than the out-painted seam would
    firsts = torch.stack([out[0, s.start] for s in segs])
    assert float(torch.pdist(firsts).min()) > 0


@pytest.mark.parametrize("ds,precision,B", [("show", "bf16", 950), ("beat", "fp32", 256)])
def test_full_batch_ddim25_loop_equals_its_rows_sampled_alone(ds, precision, B):
    """The complete ddim25 loop at the headline batch (configs[2]: SHOW B = 950 bf16, three sub-batch streams, fused-FFN path
    active; configs[1]: BEAT B = 256 fp32) with one Philox stream per clip, vs 11 of its clips sampled alone with the same
    stream: 25 compounding steps at full batch are compared with the small-batch path the goldens pin."""
    cfg = get_config(ds)
    model = gpu_model(ds, precision)
    tr = DDPMTrainer(sampler_namespace(cfg), model)
    T, Cc = cfg.n_poses, cfg.net_dim_pose
    small = make_inputs(cfg, 64, seed=19)
    rep = (B + 63) // 64
    audio = small["audio_emb"].repeat(rep, 1, 1)[:B].cuda().contiguous()
    hub = small["pretrain_aud_feat"].repeat(rep, 1, 1)[:B].cuda().contiguous()
    audio += 0.01 * torch.randn(audio.shape, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(2))
    pid = torch.zeros(B, cfg.style_dim, device="cuda:0")
    pid[torch.arange(B), torch.arange(B) % cfg.style_dim] = 1.0
    keys = list(range(1000, 1000 + B))
    full = tr.generate_batch(audio, pid, Cc, {"pretrain_aud_feat": hub}, {}, seed=77, row_keys=keys)
    assert full.shape == (B, T, Cc) and torch.isfinite(full).all()
    picks = sorted({0, 1, B // 3 - 1, B // 3, B // 3 + 1, B // 2 - 1, B // 2, 2 * B // 3 - 1, 2 * B // 3 + 1, B - 2, B - 1})   # around every possible stream split, and the batch ends
    tol = 1e-5 if precision == "fp32" else 1.2e-2
    worst = 0.0
    for b in picks:
        solo = tr.generate_batch(audio[b:b + 1], pid[b:b + 1], Cc, {"pretrain_aud_feat": hub[b:b + 1]}, {}, seed=77, row_keys=[keys[b]])
        worst = max(worst, rel_err(solo[0], full[b]))
    print(f"[full-batch ddim25 {ds} {precision} B={B}] worst rel err of {len(picks)} clips vs the clip sampled alone: {worst:.3e}")
    assert worst < tol
    # ... and the ORACLE (CPU restatement of the reference) runs the whole 25-step loop on three of those clips, started from the
    # very x_T the GPU drew for them (Philox row stream `keys[b]`, draw 0; eta = 0: no other draw enters the result), so that the
    # full-batch loop is tied to the reference's algorithm directly, not only to the small-batch GPU path (round-3 review).
    from oracle import denoiser_ref, sampler_ref
    from util import synthetic_sd
    sd = synthetic_sd(ds)
    ocl = [0, B // 2, B - 1]
    a_c, h_c, p_c = audio[ocl].cpu(), hub[ocl].cpu(), pid[ocl].cpu()
    xT = _rows(77, [keys[b] for b in ocl], T * Cc).view(len(ocl), T, Cc)

    class _Src:                      # draw 0 = x_T; the randn_like of every ddim step is multiplied by sigma = 0
        i = 0
        def randn(self, shape):
            self.i += 1
            return xT.clone() if self.i == 1 else torch.zeros(*shape)

    def eps_fn(xc, t_orig, c1, c2):
        with torch.no_grad():
            return denoiser_ref.unidiffuser(sd, cfg, xc, torch.full((len(ocl),), t_orig), c1, c2, a_c, p_c, h_c)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    xr = sampler_ref.ddim_sample_loop(eps_fn, (len(ocl), T, Cc), {}, _Src(), overlap_len=cfg.overlap_len)
    eo = max(rel_err(full[b], xr[j]) for j, b in enumerate(ocl))
    print(f"[full-batch ddim25 {ds} {precision} B={B}] clips {ocl} vs the oracle's 25-step loop from the same x_T: max err / range {eo:.3e}")
    assert eo < (1e-3 if precision == "fp32" else 1.2e-2)


@pytest.mark.parametrize("case", ["philox_plain", "philox_masked", "stack_masked", "rows_plain", "ddpm", "son_chain"])
def test_free_running_sub_batch_streams_are_bit_identical_to_one_stream(case, monkeypatch):
    """Large batches run the WHOLE sampling loop per sub-batch on its own stream (sampler.hip: one fork before the loop, one join
    after it; per-sub-batch timestep scalars, noise slices with the whole-batch Philox counters, ddim / undo / ddpm updates,
    out-painting blend).  Clips never interact, so the result must equal the single-stream run (DSH_DUAL=0) bit for bit:
    plain ddim25, the out-painting jump schedule (63 evaluations + 48 undo steps, RePaint blend), injected noise stacks,
    per-row Philox keys, the ancestral DDPM update, and a two-window --same_overlap_noisy chain (per-level noisy tails saved and
    consumed per sub-batch)."""
    from diffsheg_amd.model import UniDiffuser
    from diffsheg_amd.synthetic import SeededNoise
    from util import synthetic_sd
    cfg = get_config("show")
    B, T, Cc, L = 160, 88, cfg.net_dim_pose, cfg.overlap_len          # 14 080 token rows: two streams of 80 clips
    small = make_inputs(cfg, 16, seed=23)
    rep = lambda v: v.repeat(B // 16, *([1] * (v.dim() - 1))).contiguous()
    audio, hub = rep(small["audio_emb"]), rep(small["pretrain_aud_feat"])
    audio = audio + 0.01 * torch.arange(B, dtype=torch.float32).view(B, 1, 1)
    pid = torch.zeros(B, cfg.style_dim)
    pid[torch.arange(B), torch.arange(B) % cfg.style_dim] = 1.0
    y = {}
    if "masked" in case:
        gt = torch.zeros(B, T, Cc)
        gt[:, :L] = torch.randn(B, L, Cc, generator=torch.Generator().manual_seed(4))
        mask = torch.zeros(B, T, Cc, dtype=torch.bool)
        mask[:, :L] = True
        y = {"gt": gt, "outpainting_mask": mask}
    outs = []
    for dual in ("0", "3"):
        monkeypatch.setenv("DSH_DUAL", dual)
        model = UniDiffuser(cfg, synthetic_sd("show"), device="cuda:0", precision="bf16")
        tr = DDPMTrainer(sampler_namespace(cfg, ddim=case != "ddpm", diffusion_steps=50 if case == "ddpm" else cfg.diffusion_steps,
                                           same_overlap_noisy=case == "son_chain"), model)
        if case == "son_chain":     # --same_overlap_noisy: two windows; the second takes the first's per-level noisy tails from the context
            N = 2 * T - L
            a2 = torch.cat([audio, audio.flip(1)[:, :T - L]], 1)
            h2 = torch.cat([hub, hub.flip(1)[:, :T - L]], 1)
            out = tr.sample_arbitrary_len(a2, pid, {"pretrain_aud_feat": h2}, seed=7)
            assert out.shape == (B, N, Cc)
            outs.append(out.clone())
            del tr, model
            continue
        kw = {}
        if case.startswith("stack"):
            kw["noise_source"] = SeededNoise(321)
        else:
            kw["seed"] = 99
            if case == "rows_plain":
                kw["row_keys"] = list(range(500, 500 + B))
        out = tr.generate_batch(audio, pid, Cc, {"pretrain_aud_feat": hub}, y, **kw)
        outs.append(out.clone())
        del tr, model
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())


@pytest.mark.parametrize("mode,extra", [("chain", ["--chains", "4", "--stream-frames", "700"]), ("batch", ["--batch", "6"]),
                                        ("ddpm", ["--batch", "2"])])
def test_bench_launches_its_own_ranks(mode, extra):
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself and print n_gpus = 2 (round 2 silently
    ran one).  This box has ONE GPU: both ranks share it (DSH_BENCH_OVERSUBSCRIBE) and the barrier / max-reduce / gather go
    through gloo, because RCCL refuses two ranks on one device; everything else is the real multi-rank code path."""
    env = dict(os.environ, DSH_BENCH_OVERSUBSCRIBE="1", DSH_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", mode, "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline", "--no-roofline", "--no-chain-latency"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["mode"] == mode and d["value"] > 0 and "expected_scaling" in d


@pytest.mark.parametrize("mode,extra", [("batch", ["--batch", "16"]), ("chain", ["--chains", "16", "--stream-frames", "1500"])])
def test_bench_at_world_size_8_on_one_device(mode, extra):
    """The launcher, rendezvous, barrier + max-reduce and (chain mode) the output gather at the world size the driver's scaling run
    uses: EIGHT ranks started by `python bench.py --gpus 8`, all sharing this box's one GPU (DSH_BENCH_OVERSUBSCRIBE, gloo because
    RCCL refuses several ranks per device).  Every rank pins itself to NUMA-local cores (host_affinity in the line) and the line
    carries the host enqueue time per step."""
    env = dict(os.environ, DSH_BENCH_OVERSUBSCRIBE="1", DSH_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--mode", mode, "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline", "--no-roofline", "--no-chain-latency"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    d = json.loads(lines[0])
    print(f"[world 8, {mode}] {d['value']:.0f} frames/s, host enqueue {d['host_enqueue_ms_per_step']:.1f} ms of {d['ms_per_step']:.1f} ms per step, "
          f"affinity {d['host_affinity']}")
    assert d["n_gpus"] == 8 and d["config"]["mode"] == mode and d["value"] > 0
    assert d["host_enqueue_ms_per_step"] > 0 and "host_affinity" in d and "telemetry" in d


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def test_rccl_collectives_run_at_world_size_1():
    """The RCCL calls of the multi-GPU path — init_process_group("nccl", device_id=...), all-reduce, barrier, the device-side broadcast
    of the feature stream (broadcast_stream) and the device-side gather of the outputs (gather_outputs) — executed on this box's
    ONE GPU as a process group of one rank (DSH_FORCE_COLLECTIVES=1 keeps them from short-circuiting), and bit-identical to the
    plain call.  Before round 4 only gloo ever ran them."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_worker.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2500:])


@pytest.mark.parametrize("mode,extra", [("chain", ["--chains", "4", "--stream-frames", "700", "--inputs-on-rank0"]), ("batch", ["--batch", "6"])])
def test_bench_under_a_launcher_uses_rccl_at_one_rank(mode, extra):
    """bench.py started the way the driver starts N > 1 (python -m torch.distributed.run ... bench.py --gpus N) with N = 1: the nccl
    process group, its barrier / max-reduce and (chain mode) the broadcast + gather run over RCCL; batch mode reports rank 0's p50
    step latency, which a multi-rank line used to drop."""
    env = dict(os.environ, DSH_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--mode", mode, "--steps", "2", "--warmup", "0",
           "--no-cpu-baseline", "--no-roofline", "--no-chain-latency"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-2500:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["collective_backend"] == "nccl" and d["value"] > 0
    if mode == "batch":
        assert d["p50_step_latency_ms"] > 0
