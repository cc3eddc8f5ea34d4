"""ORACLE (test infrastructure — never imported by the product path).

CPU restatement, in plain functional PyTorch fp32, of the DiffSHEG ``UniDiffuser`` denoiser
(/root/reference/models/transformer.py).  It works directly on a state dict with the reference's
key names; no nn.Module, no mutable option namespace.  Each function cites the reference lines it
restates.  Parity pin: validated in the development container against the *imported* reference
(tests/golden/make_golden.py writes the fixtures; tests/test_oracle_golden.py replays them).  The
reference ships no tests/golden vectors of its own, so by the reference's tests alone parity is
"unpinned"; the pin is the imported-reference fixtures under tests/golden/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


def timestep_embedding(t: Tensor, dim: int) -> Tensor:
    """[cos(t f) | sin(t f)], f_j = exp(-ln(1e4) j / half)   (transformer.py:42-59)."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _lin(sd: SD, p: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


def _ln(sd: SD, p: str, x: Tensor) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def mlp_embed(sd: SD, p: str, x: Tensor) -> Tensor:
    """Linear -> SiLU -> Linear; time_embed / pid_embed (transformer.py:446-457,623-627)."""
    return _lin(sd, p + ".2", F.silu(_lin(sd, p + ".0", x)))


def stylization(sd: SD, p: str, h: Tensor, emb: Tensor) -> Tensor:
    """FiLM: LN(h)*(1+scale)+shift -> SiLU -> Linear      (transformer.py:86-97)."""
    e = _lin(sd, p + ".emb_layers.1", F.silu(emb)).unsqueeze(1)
    d = h.shape[-1]
    scale, shift = e[..., :d], e[..., d:]
    return _lin(sd, p + ".out_layers.2", F.silu(_ln(sd, p + ".norm", h) * (1 + scale) + shift))


def linear_self_attention(sd: SD, p: str, x: Tensor, emb: Tensor, n_head: int) -> Tensor:
    """Efficient attention: softmax(Q) over head channels, softmax(K) over time, A=K^T V, Y=Q A
    (transformer.py:112-130).  src_mask is all-ones at inference (transformer.py:563,579)."""
    B, T, D = x.shape
    n = _ln(sd, p + ".norm", x)
    q = _lin(sd, p + ".query", n).view(B, T, n_head, -1).softmax(dim=-1)
    k = _lin(sd, p + ".key", n).view(B, T, n_head, -1).softmax(dim=1)
    v = _lin(sd, p + ".value", n).view(B, T, n_head, -1)
    att = torch.einsum("bnhd,bnhl->bhdl", k, v)
    y = torch.einsum("bnhd,bhdl->bnhl", q, att).reshape(B, T, D)
    return x + stylization(sd, p + ".proj_out", y, emb)


def ffn(sd: SD, p: str, x: Tensor, emb: Tensor) -> Tensor:
    """Linear -> exact GELU -> Linear -> stylization -> +x   (transformer.py:178-181)."""
    y = _lin(sd, p + ".linear2", F.gelu(_lin(sd, p + ".linear1", x)))
    return x + stylization(sd, p + ".proj_out", y, emb)


def decoder_layer(sd: SD, p: str, h: Tensor, cond: Optional[Tensor], emb: Tensor, n_head: int,
                  null_cond_emb: Optional[Tensor], cfg_active: bool) -> Tensor:
    """mlp_includeX + cond_residual layer (transformer.py:300-346).

    ``cond`` is the already concatenated [audio_proj | hubert128 | (expr_x0)] block, or None for
    ``encoder_aud`` (xf=None, cond_proj=False) where the residual doubles the input."""
    if cond is not None:
        u = torch.cat((h, cond), dim=-1)
        if cfg_active:
            # eval-time CFG: first half of the doubled batch is the unconditional half and its
            # *whole* concat row (latent included) becomes null_cond_emb (transformer.py:330-332)
            nb = h.shape[0] // 2
            u = torch.cat((null_cond_emb.expand(nb, h.shape[1], -1), u[nb:]), dim=0)
        u = _ln(sd, p + ".feat_proj.0", u)
        u = _lin(sd, p + ".feat_proj.3", F.silu(_lin(sd, p + ".feat_proj.1", u)))
        h = u + h
    else:
        h = h + h
    h = linear_self_attention(sd, p + ".sa_block", h, emb, n_head)
    return ffn(sd, p + ".ffn", h, emb)


def hubert_encoder(sd: SD, p: str, hubert: Tensor) -> Tensor:
    """Conv1d(1024,128,3,p=1) -> BatchNorm1d(eval) -> GELU -> Conv1d(128,128,3,p=1) over time
    (transformer.py:437-442,515)."""
    z = F.conv1d(hubert.transpose(1, 2), sd[p + ".0.weight"], None, padding=1)
    z = F.batch_norm(z, sd[p + ".1.running_mean"], sd[p + ".1.running_var"],
                     sd[p + ".1.weight"], sd[p + ".1.bias"], False, 0.0, 1e-5)
    z = F.conv1d(F.gelu(z), sd[p + ".3.weight"], None, padding=1)
    return z.transpose(1, 2)


def motion_transformer(sd: SD, p: str, cfg, x: Tensor, t: Tensor, audio256: Tensor,
                       person_id: Tensor, hubert: Tensor, expr_cond: Optional[Tensor]) -> Tensor:
    """One MotionTransformer forward incl. CFG doubling + mix (transformer.py:496-587)."""
    if person_id.dim() == 1:
        person_id = person_id.unsqueeze(0)
    p = p + "." if p else ""                 # stand-alone MotionTransformer: no sub-module prefix
    hub = hubert_encoder(sd, p + "hubert_encoder", hubert)
    extra = hub if expr_cond is None else torch.cat((hub, expr_cond), dim=-1)
    if cfg.cfg_active:
        x, t, audio256, person_id, extra = (torch.cat([v, v]) for v in (x, t, audio256, person_id, extra))
    emb = mlp_embed(sd, p + "time_embed", timestep_embedding(t, cfg.latent_dim)) \
        + mlp_embed(sd, p + "pid_embed", person_id)
    T = x.shape[1]
    h = _lin(sd, p + "joint_embed", x) + sd[p + "PE.pe"][:, :T]
    cond = torch.cat((_lin(sd, p + "audio_proj", audio256), extra), dim=-1)
    null = sd.get(p + "null_cond_emb")
    for i in range(cfg.num_layers):
        h = decoder_layer(sd, f"{p}temporal_decoder_blocks.{i}", h, cond, emb, cfg.num_heads,
                          null, cfg.cfg_active)
    out = _lin(sd, p + "out", h)
    if cfg.cfg_active:
        nb = out.shape[0] // 2
        out = out[:nb] + cfg.cond_scale * (out[nb:] - out[:nb])
    return out


def unidiffuser(sd: SD, cfg, x: Tensor, t: Tensor, c1: Tensor, c2: Tensor, audio_emb: Tensor,
                person_id: Tensor, hubert: Tensor, return_parts: bool = False):
    """UniDiffuser.forward (transformer.py:728-770).  ``c1``/``c2`` are the broadcastable
    sqrt(1/abar_t) / sqrt(1/abar_t - 1) the sampler hands in as ``sqrt_alphas``
    (gaussian_diffusion.py:527-532)."""
    emb_a = mlp_embed(sd, "time_embed", timestep_embedding(t, cfg.latent_dim))
    aud_feat = decoder_layer(sd, "encoder_aud", audio_emb, None, emb_a, cfg.num_heads, None, False)
    audio256 = torch.cat((audio_emb, aud_feat), dim=-1)
    ges_x, exp_x = x[..., : cfg.split_pos], x[..., cfg.split_pos:]
    eps_exp = motion_transformer(sd, "encoder_exp", cfg, exp_x, t, audio256, person_id, hubert, None)
    expr_x0 = c1 * exp_x - c2 * eps_exp                       # transformer.py:717-724,749
    eps_ges = motion_transformer(sd, "encoder_ges", cfg, ges_x, t, audio256, person_id, hubert, expr_x0)
    out = torch.cat((eps_ges, eps_exp), dim=-1)
    if return_parts:
        return out, {"aud_feat": aud_feat, "eps_exp": eps_exp, "expr_x0": expr_x0, "eps_ges": eps_ges}
    return out


def single_motion_transformer(sd: SD, cfg, x: Tensor, t: Tensor, audio_emb: Tensor, person_id: Tensor, hubert: Tensor) -> Tensor:
    """The model runner.py:46-57 builds with ``opt.unidiffuser = False`` (model_base 'transformer_encoder'): one
    MotionTransformer over all gesture | expression channels; ``audio_proj`` reads the 128 mel features directly and there is
    no encoder_aud / expression -> gesture flow (transformer.py:496-587; the sampler passes no sqrt_alphas,
    gaussian_diffusion.py:527-536)."""
    return motion_transformer(sd, "", cfg, x, t, audio_emb, person_id, hubert, None)
