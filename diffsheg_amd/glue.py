"""Rows either side of the sampling path (SURVEY.md §8f "next"), to the same drop-in standard:

* :func:`load_checkpoint` — the reference's ``.tar`` (``torch.save`` dict, weights under ``['encoder']``,
  DDP ``module.`` prefix tolerated, ``strict=False``; /root/reference/trainers/ddpm_show_trainer.py:259-292);
* :func:`interpolate_features` — HuBERT hidden states resampled to the pose frame rate with
  ``F.interpolate(mode='linear', align_corners=True)`` (datasets/show.py:98, ddpm_show_trainer.py:1082), HIP kernel;
* :func:`inv_standardize` / :func:`split_motion` — de-normalisation and gesture|expression split of the sampled
  window chain (datasets/show.py:157-162, ddpm_show_trainer.py:906-921), HIP kernel, output stays on the device.

HuBERT itself (``facebook/hubert-large-ls960-ft``) and mel extraction stay third-party and out of scope.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import torch

from . import _lib
from .weights import strip_ddp_prefix


def load_checkpoint(path: str) -> Tuple[Dict[str, torch.Tensor], Dict[str, object]]:
    """Return (UniDiffuser state dict, metadata) from a reference checkpoint file."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    if "encoder" not in ckpt:
        raise KeyError("checkpoint has no 'encoder' entry (expected the dict written by DDPMTrainer_*.save)")
    meta = {k: ckpt.get(k) for k in ("ep", "total_it", "best_fgd", "best_mse", "best_pck") if k in ckpt}
    return strip_ddp_prefix(ckpt["encoder"]), meta


def _stream_ptr(device: torch.device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def interpolate_features(feat: torch.Tensor, frames_out: int) -> torch.Tensor:
    """[B, T_in, C] (or [T_in, C]) fp32 on a GPU -> [B, frames_out, C], linear with align_corners=True."""
    if not feat.is_cuda:
        raise _lib.DshError("interpolate_features runs on the GPU (no CPU fallback)")
    squeeze = feat.dim() == 2
    x = (feat.unsqueeze(0) if squeeze else feat).to(torch.float32).contiguous()
    B, Tin, Cc = x.shape
    y = torch.empty(B, frames_out, Cc, device=x.device)
    _lib.check(_lib.lib().dsh_interp_time(_stream_ptr(x.device), x.data_ptr(), B, Tin, Cc, y.data_ptr(), frames_out),
               "dsh_interp_time")
    return y.squeeze(0) if squeeze else y


def inv_standardize(motion: torch.Tensor, mean: torch.Tensor, std: torch.Tensor) -> torch.Tensor:
    """``motion * std + mean`` over the channel (last) axis, on the device."""
    if not motion.is_cuda:
        raise _lib.DshError("inv_standardize runs on the GPU (no CPU fallback)")
    x = motion.to(torch.float32).contiguous()
    Cc = x.shape[-1]
    m = mean.to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
    s = std.to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
    if m.numel() != Cc or s.numel() != Cc:
        raise ValueError(f"mean/std must have {Cc} entries")
    y = torch.empty_like(x)
    _lib.check(_lib.lib().dsh_inv_standardize(_stream_ptr(x.device), x.data_ptr(), x.numel(), Cc, m.data_ptr(), s.data_ptr(),
                                              y.data_ptr()), "dsh_inv_standardize")
    return y


def split_motion(motion: torch.Tensor, split_pos: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """gesture | expression split of the joint motion tensor (ddpm_show_trainer.py:920-921)."""
    return motion[..., :split_pos], motion[..., split_pos:]
