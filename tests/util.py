"""Shared helpers for the parity tests."""
from __future__ import annotations

import ctypes as C
import functools
import os

import numpy as np
import torch

from diffsheg_amd.config import get_config
from diffsheg_amd.weights import make_synthetic_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
WEIGHT_SEED = 1234


def golden(name: str):
    return np.load(os.path.join(GOLDEN, name))


@functools.lru_cache(maxsize=4)
def synthetic_sd(ds: str):
    return make_synthetic_state_dict(get_config(ds), WEIGHT_SEED)


_MODELS = {}


def gpu_model(ds: str, precision: str = "fp32", **cfg_over):
    """One UniDiffuser handle per (dataset, precision, overrides) for the whole test session."""
    from diffsheg_amd.model import UniDiffuser
    key = (ds, precision, tuple(sorted(cfg_over.items())))
    if key not in _MODELS:
        cfg = get_config(ds, **cfg_over)
        _MODELS[key] = UniDiffuser(cfg, synthetic_sd(ds), device="cuda:0", precision=precision)
    return _MODELS[key]


def gpu_single_model(precision: str = "fp32", ds: str = "show"):
    """The opt.unidiffuser = False model (runner.py:46-57): one MotionTransformer over all channels."""
    from diffsheg_amd.model import MotionTransformer
    key = (ds, precision, "single")
    if key not in _MODELS:
        cfg = get_config(ds, unidiffuser=False)
        _MODELS[key] = MotionTransformer(cfg, make_synthetic_state_dict(cfg, WEIGHT_SEED), device="cuda:0", precision=precision)
    return _MODELS[key]


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def max_abs(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())
