"""Weight naming + a seeded synthetic weight generator.

Key names and shapes mirror the reference ``UniDiffuser.state_dict()`` (SURVEY.md §8b-4;
/root/reference/models/transformer.py:590-700) so a real ``checkpoint['encoder']``
(/root/reference/trainers/ddpm_show_trainer.py:259-292) loads unchanged.  No trained checkpoint
is reachable offline, so tests / bench use :func:`make_synthetic_state_dict`: every tensor is
drawn from its own ``torch.Generator`` seeded by (seed, key-name) — order independent, identical
on the fixture-generation side (loaded into the imported reference) and on the GPU box.

The reference zero-initialises ``*.proj_out.out_layers.2`` and ``*.ffn.linear2``
(transformer.py:62-68,83,173); a random-init model is therefore an identity through every
attention / FFN / stylization block.  The synthetic generator gives those tensors N(0, 0.02²)
values and randomises LayerNorm affine + BatchNorm running stats so every kernel is exercised.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, List, Tuple

import torch

from .config import DiffSHEGConfig

Spec = Tuple[str, Tuple[int, ...], str]   # (key, shape, kind)


def _linear(prefix: str, out_f: int, in_f: int, kind: str = "linear") -> List[Spec]:
    return [(f"{prefix}.weight", (out_f, in_f), kind), (f"{prefix}.bias", (out_f,), kind + "_bias")]


def _layernorm(prefix: str, d: int) -> List[Spec]:
    return [(f"{prefix}.weight", (d,), "ln_w"), (f"{prefix}.bias", (d,), "ln_b")]


def _stylization(prefix: str, d: int, e: int) -> List[Spec]:
    return (_linear(f"{prefix}.emb_layers.1", 2 * d, e) + _layernorm(f"{prefix}.norm", d)
            + _linear(f"{prefix}.out_layers.2", d, d, kind="zero_linear"))


def _layer(prefix: str, d: int, e: int, ff: int, concat_dim: int | None) -> List[Spec]:
    s: List[Spec] = []
    if concat_dim is not None:
        s += _layernorm(f"{prefix}.feat_proj.0", concat_dim)
        s += _linear(f"{prefix}.feat_proj.1", 2 * d, concat_dim)
        s += _linear(f"{prefix}.feat_proj.3", d, 2 * d)
    s += _layernorm(f"{prefix}.sa_block.norm", d)
    for n in ("query", "key", "value"):
        s += _linear(f"{prefix}.sa_block.{n}", d, d)
    s += _stylization(f"{prefix}.sa_block.proj_out", d, e)
    s += _linear(f"{prefix}.ffn.linear1", ff, d)
    s += _linear(f"{prefix}.ffn.linear2", d, ff, kind="zero_linear")
    s += _stylization(f"{prefix}.ffn.proj_out", d, e)
    return s


def _motion_transformer(prefix: str, cfg: DiffSHEGConfig, in_feats: int, concat_dim: int, audio_in: int) -> List[Spec]:
    d, e = cfg.latent_dim, cfg.time_embed_dim
    prefix = prefix + "." if prefix else ""            # the stand-alone MotionTransformer has no sub-module prefix
    s: List[Spec] = []
    if cfg.classifier_free:
        s.append((f"{prefix}null_cond_emb", (1, concat_dim), "normal1"))
    s.append((f"{prefix}PE.pe", (1, cfg.pe_max_len, d), "pe"))
    s += _linear(f"{prefix}joint_embed", d, in_feats)
    s += _linear(f"{prefix}audio_proj", cfg.aud_latent_dim, audio_in)
    s.append((f"{prefix}hubert_encoder.0.weight", (cfg.hubert_enc_dim, cfg.hubert_dim, 3), "conv"))
    s.append((f"{prefix}hubert_encoder.1.weight", (cfg.hubert_enc_dim,), "ln_w"))
    s.append((f"{prefix}hubert_encoder.1.bias", (cfg.hubert_enc_dim,), "ln_b"))
    s.append((f"{prefix}hubert_encoder.1.running_mean", (cfg.hubert_enc_dim,), "bn_mean"))
    s.append((f"{prefix}hubert_encoder.1.running_var", (cfg.hubert_enc_dim,), "bn_var"))
    s.append((f"{prefix}hubert_encoder.1.num_batches_tracked", (), "counter"))
    s.append((f"{prefix}hubert_encoder.3.weight", (cfg.hubert_enc_dim, cfg.hubert_enc_dim, 3), "conv"))
    s += _linear(f"{prefix}time_embed.0", e, d) + _linear(f"{prefix}time_embed.2", e, e)
    s += _linear(f"{prefix}pid_embed.0", e, cfg.style_dim) + _linear(f"{prefix}pid_embed.2", e, e)
    for i in range(cfg.num_layers):
        s += _layer(f"{prefix}temporal_decoder_blocks.{i}", d, e, cfg.ff_size, concat_dim)
    s += _linear(f"{prefix}out", in_feats, d)
    return s


def state_dict_spec(cfg: DiffSHEGConfig) -> List[Spec]:
    """All ``UniDiffuser`` (or, with ``cfg.unidiffuser == False``, stand-alone ``MotionTransformer``: runner.py:46-57)
    state-dict entries, in the reference's registration order."""
    d, e = cfg.latent_dim, cfg.time_embed_dim
    if not cfg.unidiffuser:
        return _motion_transformer("", cfg, cfg.net_dim_pose, cfg.concat_dim_single, cfg.audio_dim)
    s: List[Spec] = []
    s += _linear("time_embed.0", e, d) + _linear("time_embed.2", e, e)
    # encoder_aud: one layer at D = audio_dim with cond_proj=False (transformer.py:629-640)
    s += _layer("encoder_aud", cfg.audio_dim, e, cfg.ff_size, None)
    s += _motion_transformer("encoder_exp", cfg, cfg.expression_dim, cfg.concat_dim_exp, 2 * cfg.audio_dim)
    s += _motion_transformer("encoder_ges", cfg, cfg.dim_pose, cfg.concat_dim_ges, 2 * cfg.audio_dim)
    return s


def positional_table(cfg: DiffSHEGConfig) -> torch.Tensor:
    """``PeriodicPositionalEncoding(period=600)`` buffer (transformer.py:19-31): sin on even
    channels, cos on odd, frame index taken mod ``period``, repeated to 1200 rows."""
    d, period = cfg.latent_dim, cfg.pe_period
    pos = torch.arange(period, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2).float() * (-math.log(10000.0) / d))
    pe = torch.zeros(period, d)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    reps = cfg.pe_max_len // period          # reference: max_seq_len(600)//period + 1 = 2
    return pe.repeat(reps, 1).unsqueeze(0)


def _gen(seed: int, key: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1000003 + zlib.crc32(key.encode())) & 0x7FFFFFFFFFFFFFFF)
    return g


def make_synthetic_state_dict(cfg: DiffSHEGConfig, seed: int = 1234,
                              out_scale: float = 1.0) -> "OrderedDict[str, torch.Tensor]":
    """Seeded fp32 CPU state dict with the reference's keys/shapes (see module docstring)."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for key, shape, kind in state_dict_spec(cfg):
        g = _gen(seed, key)
        if kind in ("linear", "conv"):
            fan_in = 1
            for s_ in shape[1:]:
                fan_in *= s_
            bound = 1.0 / math.sqrt(fan_in)
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif kind == "linear_bias":
            t = (torch.rand(shape, generator=g) * 2 - 1) * 0.05
        elif kind == "zero_linear":
            t = torch.randn(shape, generator=g) * 0.02
        elif kind == "zero_linear_bias":
            t = torch.randn(shape, generator=g) * 0.02
        elif kind == "ln_w":
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif kind in ("ln_b", "bn_mean"):
            t = 0.1 * torch.randn(shape, generator=g)
        elif kind == "bn_var":
            t = 0.5 + torch.rand(shape, generator=g)
        elif kind == "normal1":
            t = torch.randn(shape, generator=g)
        elif kind == "pe":
            t = positional_table(cfg)
            assert tuple(t.shape) == shape, (t.shape, shape)
        elif kind == "counter":
            t = torch.zeros((), dtype=torch.int64)
        else:  # pragma: no cover
            raise AssertionError(kind)
        if out_scale != 1.0 and key.endswith(".out.weight"):
            t = t * out_scale
        sd[key] = t.contiguous()
    return sd


def validate_state_dict(cfg: DiffSHEGConfig, sd: Dict[str, torch.Tensor]) -> None:
    """Raise ``KeyError`` / ``ValueError`` if ``sd`` is not a complete UniDiffuser state dict.

    DDP checkpoints carry a ``module.`` prefix (ddpm_show_trainer.py:278-292); callers strip it
    with :func:`strip_ddp_prefix` first.
    """
    for key, shape, kind in state_dict_spec(cfg):
        if key not in sd:
            if kind in ("counter", "pe"):
                continue
            raise KeyError(f"missing weight {key!r}")
        if tuple(sd[key].shape) != tuple(shape):
            raise ValueError(f"weight {key!r}: shape {tuple(sd[key].shape)} != expected {shape}")


def strip_ddp_prefix(sd: Dict[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in sd.items())


def make_cross_attention_state_dict(seed: int, latent_dim: int = 512, aud_latent_dim: int = 256, time_embed_dim: int = 2048) -> Dict[str, torch.Tensor]:
    """Seeded synthetic parameters of a ``LinearTemporalCrossAttention`` (models/transformer.py:133-145) under the module's
    own parameter names; used by the golden generator (loaded into the reference module) and by the parity test."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    D, L, E = latent_dim, aud_latent_dim, time_embed_dim

    def lin(n, k, scale=None):
        return torch.randn(n, k, generator=g) * (scale if scale is not None else k ** -0.5), 0.1 * torch.randn(n, generator=g)

    sd: Dict[str, torch.Tensor] = {}
    for name, dim in (("norm", D), ("text_norm", L), ("proj_out.norm", D)):
        sd[name + ".weight"] = 1 + 0.1 * torch.randn(dim, generator=g)
        sd[name + ".bias"] = 0.1 * torch.randn(dim, generator=g)
    for name, (n, k) in (("query", (D, D)), ("key", (D, L)), ("value", (D, L)), ("proj_out.emb_layers.1", (2 * D, E)),
                         ("proj_out.out_layers.2", (D, D))):
        sd[name + ".weight"], sd[name + ".bias"] = lin(n, k)
    return sd
