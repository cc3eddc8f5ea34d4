"""Seeded synthetic conditioning.  There are no datasets / HuBERT weights offline,
so tests, fixtures and bench all draw inputs of the reference's shapes from CPU generators:

    audio_emb (mel)            ~ N(0,1)  [B,T,128]      datasets/show.py:65-144
    pretrain_aud_feat (HuBERT) ~ N(0,1)  [B,T,1024]
    person_id                  one-hot   [B,S]   row i -> i mod S
    x_T                        ~ N(0,1)  [B,T,C]

CPU ``torch.Generator`` streams are bit-reproducible across machines running the same torch
build, so fixtures store only seeds + expected outputs.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .config import DiffSHEGConfig

_TORCH_RANDN = torch.randn     # bound at import: the fixture generator monkey-patches torch.randn


def make_inputs(cfg: DiffSHEGConfig, batch: int, frames: Optional[int] = None, seed: int = 3) -> Dict[str, torch.Tensor]:
    T = cfg.n_poses if frames is None else frames
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    audio = torch.randn(batch, T, cfg.audio_dim, generator=g)
    hubert = torch.randn(batch, T, cfg.hubert_dim, generator=g)
    x_t = torch.randn(batch, T, cfg.net_dim_pose, generator=g)
    pid = torch.zeros(batch, cfg.style_dim)
    pid[torch.arange(batch), torch.arange(batch) % cfg.style_dim] = 1.0
    return {"audio_emb": audio, "pretrain_aud_feat": hubert, "person_id": pid, "x_T": x_t}


class SeededNoise:
    """Gaussian draws in the reference's draw order from a seeded CPU generator.

    ``randn(shape)`` returns a CPU fp32 tensor; the product sampler uploads it, the fixture
    generator monkey-patches the reference's ``th.randn`` / ``th.randn_like`` with it.
    """

    def __init__(self, seed: int):
        self.gen = torch.Generator(device="cpu")
        self.gen.manual_seed(seed)
        self.count = 0

    def randn(self, shape) -> torch.Tensor:
        self.count += 1
        return _TORCH_RANDN(*tuple(shape), generator=self.gen)
