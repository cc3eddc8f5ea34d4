"""Harness boundary: ``generate_batch`` and the arbitrary-length
window chain of ``test_arbitrary_len`` / ``test_custom_aud``
(/root/reference/trainers/ddpm_show_trainer.py:163-198, :801-819, :864-906; BEAT twin
ddpm_beat_trainer.py:185-220, :932-1039), plus the sharding of independent chains over ranks
(the reference shards test videos with a DistributedSampler, ddpm_show_trainer.py:743-750).

Training, metrics, BVH/JSON writers, HuBERT extraction and checkpoint I/O are out of scope.
"""
from __future__ import annotations

import os

import argparse
from typing import Dict, List, Optional, Sequence

import torch

from .config import DiffSHEGConfig
from .diffusion import (GaussianDiffusion, ModelMeanType, ModelVarType, SpacedDiffusion, get_named_beta_schedule,
                        space_timesteps)
from .model import UniDiffuser


def sampler_namespace(cfg: DiffSHEGConfig, **over) -> argparse.Namespace:
    """The `opt` attributes the sampler reads (gaussian_diffusion.py / scheduler.py)."""
    ns = argparse.Namespace(jump_length=cfg.jump_length, jump_n_sample=cfg.jump_n_sample, overlap_len=cfg.overlap_len,
                            addBlend=cfg.add_blend, no_resample=cfg.no_resample, no_repaint=cfg.no_repaint,
                            timestep_respacing=cfg.timestep_respacing, unidiffuser=cfg.unidiffuser, same_overlap_noisy=False,
                            fix_head_var=False, ddim=True, n_poses=cfg.n_poses, net_dim_pose=cfg.net_dim_pose,
                            PE="pe_sinu", diffusion_steps=cfg.diffusion_steps, fix_very_first=False,
                            dataset_name="talkshow" if cfg.dataset == "show" else "beat")
    for k, v in over.items():
        setattr(ns, k, v)
    return ns


_M64 = (1 << 64) - 1


def _splitmix64(z: int) -> int:
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def window_seed(seed: int, window: int) -> int:
    """Philox key of window ``window`` of a chain sampled with base seed ``seed``: a 64-bit hash of the pair, so that
    neighbouring (seed, window) pairs share no stream (``seed + window`` collides for (s, w + 1) and (s + 1, w))."""
    return _splitmix64(_splitmix64(int(seed) & _M64) ^ ((int(window) + 1) * 0xD1342543DE82EF95 & _M64))


def get_windows(x, size: int, step: int):
    """ddpm_show_trainer.py:801-819 — tensors or dicts of tensors; the tail window may be shorter."""
    if isinstance(x, dict):
        per_key = {k: get_windows(v, size, step) for k, v in x.items()}
        n = len(next(iter(per_key.values())))
        return [{k: per_key[k][i] for k in per_key} for i in range(n)]
    seq_len = x.shape[1]
    if seq_len <= size:
        return [x]
    win_num = (seq_len - (size - step)) / float(step)
    out = [x[:, m * step: m * step + size, ...] for m in range(int(win_num))]
    if win_num - int(win_num) != 0:
        out.append(x[:, int(win_num) * step:, ...])
    return out


class DDPMTrainer:
    """Sampling half of DDPMTrainer_show / DDPMTrainer_beat (ddpm_show_trainer.py:40-90)."""

    def __init__(self, args, encoder: UniDiffuser, eval_model=None):
        self.opt = args
        self.encoder = encoder
        self.device = encoder.device
        self.diffusion_steps = int(getattr(args, "diffusion_steps", 1000))
        betas = get_named_beta_schedule("linear", self.diffusion_steps)
        kw = dict(opt=args, betas=betas, model_mean_type=ModelMeanType.EPSILON,
                  model_var_type=ModelVarType.FIXED_SMALL, loss_type=None)
        self.diffusion = GaussianDiffusion(**kw)
        # the reference hard-codes 'ddim25' here (ddpm_show_trainer.py:73, ddpm_beat_trainer.py:76)
        self.diffusion_ddim_val = SpacedDiffusion(use_timesteps=space_timesteps(self.diffusion_steps, "ddim25"),
                                                  rescale_timesteps=False, **kw)

    def generate_batch(self, audio_emb, p_id, dim_pose, add_cond={}, inpaint_dict=None, **sampler_kw):
        """ddpm_show_trainer.py:163-198.  ``sampler_kw`` (noise_source= / seed=) is this build's
        noise-injection hook; the reference draws from the global torch RNG."""
        audio_emb = audio_emb.to(self.device)
        B, T = len(audio_emb), audio_emb.shape[1]
        cur_len = torch.full((B,), T, dtype=torch.long, device=self.device)
        model_kwargs = {"audio_emb": audio_emb, "length": cur_len, "person_id": p_id, "add_cond": add_cond,
                        "y": inpaint_dict, "pe_type": getattr(self.opt, "PE", "pe_sinu")}
        if getattr(self.opt, "ddim", True):
            return self.diffusion_ddim_val.ddim_sample_loop(self.encoder, (B, T, dim_pose), clip_denoised=False,
                                                            progress=True, model_kwargs=model_kwargs, **sampler_kw)
        return self.diffusion.p_sample_loop(self.encoder, (B, T, dim_pose), clip_denoised=False, progress=True,
                                            model_kwargs=model_kwargs, **sampler_kw)

    # ---- H2: arbitrary-length chain ---------------------------------------------------------
    def sample_arbitrary_len(self, audio_emb: torch.Tensor, p_id: torch.Tensor, add_cond: Dict[str, torch.Tensor],
                             noise_source_for_window=None, seed: Optional[int] = None,
                             motions: Optional[torch.Tensor] = None, row_keys: Optional[Sequence[int]] = None) -> torch.Tensor:
        """The per-video body of test_arbitrary_len (ddpm_show_trainer.py:864-906): windows of n_poses
        with stride n_poses-overlap_len; window k>0 out-paints from the last overlap_len frames of
        window k-1 (sequential chain).  Output stays on the device (the reference copies every window
        to the host).  ``opt.fix_very_first`` (ddpm_show_trainer.py:885-888): window 0 is out-painted too, from the
        LAST overlap_len frames of the first ground-truth window of ``motions`` (standardised, [B, N, C]) — the
        reference's indexing, kept as is.  Batch rows are independent chains of equal length."""
        opt = self.opt
        n_poses, L, C = int(opt.n_poses), int(opt.overlap_len), int(opt.net_dim_pose)
        step = n_poses - L
        audio_list = get_windows(audio_emb, n_poses, step)
        cond_list = get_windows(add_cond, n_poses, step) if add_cond not in (None, {}) else [{}] * len(audio_list)
        fix_first = bool(getattr(opt, "fix_very_first", False)) and L > 0
        if fix_first and motions is None:
            raise ValueError("fix_very_first needs the ground-truth motions of the clip")
        motion_list = get_windows(motions.to(self.device), n_poses, step) if fix_first else None
        outs: List[torch.Tensor] = []
        outputs = None
        son = bool(getattr(opt, "same_overlap_noisy", False))     # ddpm_beat_trainer.py:1006,1022-1028
        previous_noisy_tail = None
        for ii, (a, cnd) in enumerate(zip(audio_list, cond_list)):
            inpaint_dict = {"clip_idx": ii} if son else {}
            if L > 0:
                B, T = a.shape[0], a.shape[1]
                inpaint_dict["gt"] = torch.zeros(B, T, C, device=self.device)
                inpaint_dict["outpainting_mask"] = torch.zeros(B, T, C, dtype=torch.bool, device=self.device)
                inpaint_dict["outpainting_mask_any"] = ii > 0 or fix_first     # what `True in mask` would say, without the sync
                if ii == 0 and fix_first:
                    inpaint_dict["outpainting_mask"][..., :L, :] = True
                    inpaint_dict["gt"][:, :L, ...] = motion_list[0][:, -L:, ...]
                elif ii > 0:
                    inpaint_dict["outpainting_mask"][..., :L, :] = True
                    inpaint_dict["gt"][:, :L, ...] = outputs[:, -L:, ...]
                    if son:
                        inpaint_dict["previous_noisy_tail"] = previous_noisy_tail
            kw = {}
            if noise_source_for_window is not None:
                kw["noise_source"] = noise_source_for_window(ii)
            elif seed is not None:
                kw["seed"] = window_seed(seed, ii)
            if row_keys is not None:
                kw["row_keys"] = row_keys          # Philox: one stream per (window, chain): key = hash(seed, window), counter high words = chain id
            outputs = self.generate_batch(a, p_id, C, cnd, inpaint_dict, **kw)
            if son:
                previous_noisy_tail, outputs = outputs["saved_noisy_tail"], outputs["sample"]
            outs.append(outputs if ii == len(audio_list) - 1 else outputs[:, :step])
        return torch.cat(outs, dim=1)


    def sample_arbitrary_len_sharded(self, *args, **kw) -> Optional[torch.Tensor]:
        """Long stream -> independent chains over the ranks -> gather on rank 0 (module-level function below)."""
        return sample_arbitrary_len_sharded(self, *args, **kw)


# ---- multi-GPU: independent chains / batch rows sharded over ranks -----------------
def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced split of ``n_items`` independent units; no data-path collective is needed
    because no tensor couples two batch rows / two chains anywhere on the path."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def split_segments(n_frames: int, n_segments: int, n_poses: int, overlap_len: int) -> List[range]:
    """Cut a long feature stream into ``n_segments`` contiguous, independently-chained segments whose
    lengths are whole numbers of window strides where possible.  Seams between segments are not
    out-painted (each segment starts with an un-masked window), exactly like separate test videos."""
    step = n_poses - overlap_len
    n_strides = max(1, (n_frames - overlap_len) // step) if n_frames > n_poses else 1
    n_segments = max(1, min(n_segments, n_strides))
    segs, start = [], 0
    for i in range(n_segments):
        strides = n_strides // n_segments + (1 if i < n_strides % n_segments else 0)
        end = n_frames if i == n_segments - 1 else start + strides * step
        segs.append(range(start, end))
        start = end
    return segs


def _collectives_active(group=None) -> bool:
    """True when the per-rank code paths (broadcast / shard / gather) must run: more than one rank, or — DSH_FORCE_COLLECTIVES=1 —
    an initialised group of ONE rank.  The latter exists so that the RCCL calls (broadcast, gather on device buffers) execute on a
    single-GPU test box instead of short-circuiting at world size 1 (tests/test_gpu_sharded.py)."""
    import torch.distributed as dist
    if not dist.is_initialized():
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("DSH_FORCE_COLLECTIVES") == "1"


def broadcast_stream(t: Optional[torch.Tensor], device, src: int = 0, group=None) -> torch.Tensor:
    """rank ``src`` -> all: one conditioning tensor (mel [1,N,128] / HuBERT [1,N,1024]; 41 MB for 5 min of audio)."""
    import torch.distributed as dist
    if not _collectives_active(group):
        return t
    rank = dist.get_rank(group)
    meta = torch.zeros(8, dtype=torch.long, device=device)
    if rank == src:
        meta[0] = t.dim()
        meta[1:1 + t.dim()] = torch.tensor(t.shape, device=device)
    dist.broadcast(meta, src, group=group)
    shape = [int(v) for v in meta[1:1 + int(meta[0])]]
    buf = t.to(device=device, dtype=torch.float32).contiguous() if rank == src else torch.empty(shape, device=device)
    dist.broadcast(buf, src, group=group)
    return buf


def sample_arbitrary_len_sharded(trainer: "DDPMTrainer", audio_emb: Optional[torch.Tensor], p_id: torch.Tensor,
                                 add_cond: Optional[Dict[str, torch.Tensor]], n_segments: int, seed: int = 0, group=None,
                                 inputs_on_rank0_only: bool = False, max_chains_per_batch: int = 64) -> Optional[torch.Tensor]:
    """BASELINE config 4: one long feature stream ``[1, N, ...]`` sampled on all ranks of ``group``.

    Windows of ONE chain are sequential (window k needs the final sample of window k-1 at every denoising step,
    ddpm_show_trainer.py:891-893), so the stream is cut into ``n_segments`` independent chains
    (:func:`split_segments`; seams are not out-painted, exactly like separate test videos, which is how the reference
    itself parallelises: DistributedSampler over videos, one chain per rank, ddpm_show_trainer.py:743-750,924-931).
    Each rank owns a contiguous run of segments (:func:`shard_range`), samples equally long ones together as a batched
    chain (batch row = chain), and rank 0 gathers the frames (RCCL gather, 8.4 MB for 9000 frames).  There is no other
    collective on the data path.  Noise: on-device Philox, key = hash(seed, window index), counter high words = segment id, so every chain is
    sampled identically whatever the world size or batching.  Returns ``[1, N, C]`` on rank 0, ``None`` elsewhere.
    """
    import torch.distributed as dist
    opt = trainer.opt
    n_poses, L, C = int(opt.n_poses), int(opt.overlap_len), int(opt.net_dim_pose)
    dev = trainer.device
    ddp = _collectives_active(group)
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if ddp else (0, 1)
    add_cond = add_cond or {}
    if inputs_on_rank0_only and ddp:
        keys = [sorted(add_cond.keys()) if rank == 0 else None]
        dist.broadcast_object_list(keys, 0, group=group)
        audio_emb = broadcast_stream(audio_emb, dev, 0, group)
        add_cond = {k: broadcast_stream(add_cond[k] if rank == 0 else None, dev, 0, group) for k in keys[0]}
    if audio_emb.shape[0] != 1:
        raise ValueError("sample_arbitrary_len_sharded takes ONE stream [1, N, ...]; batch several streams by calling it per stream")
    N = int(audio_emb.shape[1])
    segs = split_segments(N, n_segments, n_poses, L)
    mine = shard_range(len(segs), rank, world)
    pid = p_id if p_id.dim() == 2 else p_id.unsqueeze(0)
    by_len: Dict[int, List[int]] = {}
    for si in mine:
        by_len.setdefault(len(segs[si]), []).append(si)
    local: Dict[int, torch.Tensor] = {}
    for ids in by_len.values():
        for c0 in range(0, len(ids), max_chains_per_batch):
            chunk = ids[c0:c0 + max_chains_per_batch]
            a = torch.cat([audio_emb[:, segs[i].start:segs[i].stop] for i in chunk], 0)
            cnd = {k: torch.cat([v[:, segs[i].start:segs[i].stop] for i in chunk], 0) for k, v in add_cond.items()}
            out = trainer.sample_arbitrary_len(a, pid[:1].expand(len(chunk), -1), cnd, seed=seed, row_keys=chunk)
            for j, i in enumerate(chunk):
                local[i] = out[j]
    loc = torch.cat([local[i] for i in mine], 0) if len(mine) else torch.zeros(0, C, device=dev)
    sizes = [sum(len(segs[i]) for i in shard_range(len(segs), r, world)) for r in range(world)]
    parts = gather_outputs(loc, sizes, group)
    if parts is None:
        return None
    return torch.cat(parts, 0).unsqueeze(0)


def gather_outputs(local: torch.Tensor, world_sizes: Sequence[int], group=None) -> Optional[List[torch.Tensor]]:
    """all ranks -> rank 0 gather of per-rank outputs with differing leading dims (RCCL / gloo)."""
    import torch.distributed as dist
    if not _collectives_active(group):
        return [local]
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mx = max(world_sizes)
    mx = max(mx, 1)
    # (a gloo group gathers through the host: single-GPU test boxes that oversubscribe one device cannot use RCCL)
    via = local.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=via)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, bufs, dst=0, group=group)
    if rank != 0:
        return None
    return [b[:n].to(local.device) for b, n in zip(bufs, world_sizes)]
