"""ORACLE (test infrastructure — never imported by the product path).

CPU restatement of the DiffSHEG sampling loops: coefficient tables, ddimN respacing, the RePaint
jump schedule, the DDIM / harmonize / DDPM loops and the sliding-window chain.  Follows
/root/reference/models/gaussian_diffusion.py, respace.py, scheduler.py and
trainers/ddpm_show_trainer.py (line cites on each function).  Pinned by fixtures generated from
the imported reference (tests/golden/make_golden.py); the reference has no tests of its own.

The denoiser is passed in as ``eps_fn(x, t_orig, c1, c2) -> eps`` so the same loops can drive the
oracle denoiser (oracle.denoiser_ref) for the CPU baseline.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

Tensor = torch.Tensor


# ----------------------------------------------------------------------------- tables
def linear_betas(n: int) -> np.ndarray:
    """gaussian_diffusion.py:234-251 ('linear' schedule, float64)."""
    scale = 1000.0 / n
    return np.linspace(scale * 1e-4, scale * 0.02, n, dtype=np.float64)


def diffusion_tables(betas: np.ndarray) -> Dict[str, np.ndarray]:
    """All float64 coefficient tables of GaussianDiffusion.__init__ (gaussian_diffusion.py:334-390)."""
    betas = np.asarray(betas, dtype=np.float64)
    ac = np.cumprod(1.0 - betas)
    ac_prev = np.append(1.0, ac[:-1])
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    return {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / ac - 1.0),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": np.log(np.append(post_var[1], post_var[1:])),
        "posterior_mean_coef1": betas * np.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * np.sqrt(1.0 - betas) / (1.0 - ac),
    }


def ddim_timestep_map(n: int, spec: str) -> List[int]:
    """'ddimK': first integer stride giving exactly K kept steps (respace.py:25-33)."""
    k = int(spec[len("ddim"):])
    for stride in range(1, n):
        if len(range(0, n, stride)) == k:
            return list(range(0, n, stride))
    raise ValueError(f"cannot create exactly {k} steps with an integer stride")


def spaced_tables(n: int, spec: str):
    """SpacedDiffusion: new betas from kept cumulative alphas (respace.py:68-82)."""
    base = diffusion_tables(linear_betas(n))
    tmap = ddim_timestep_map(n, spec)
    last, nb = 1.0, []
    for i in tmap:
        nb.append(1.0 - base["alphas_cumprod"][i] / last)
        last = base["alphas_cumprod"][i]
    return diffusion_tables(np.array(nb)), tmap


def jump_schedule(respacing: int = 25, jump_length: int = 1, jump_n_sample: int = 1) -> List[int]:
    """RePaint-style level list used by the DDIM harmonize loop (scheduler.py:178-208)."""
    t_T = 15 if respacing == 25 else int(respacing * 0.6)
    jumps = {j: jump_n_sample - 1 for j in range(0, t_T - jump_length, jump_length)}
    t, ts = t_T, []
    while t >= 1:
        t -= 1
        ts.append(t)
        if jumps.get(t, 0) > 0:
            jumps[t] -= 1
            for _ in range(jump_length):
                t += 1
                ts.append(t)
    ts.append(-1)
    return ts


def _f32(a: np.ndarray, k: int) -> Tensor:
    """fp64 table entry cast to fp32 at gather time (gaussian_diffusion.py:1514)."""
    return torch.tensor(np.float32(a[k]))


# ----------------------------------------------------------------------------- noise plumbing
class NoiseSource:
    """Gaussian draws in the reference's order.  Either replays a recorded
    stack or draws from a seeded CPU generator."""

    def __init__(self, stack: Optional[Sequence[Tensor]] = None, seed: Optional[int] = None):
        self.stack = list(stack) if stack is not None else None
        self.i = 0
        self.gen = None
        if stack is None:
            self.gen = torch.Generator(device="cpu")
            self.gen.manual_seed(0 if seed is None else seed)

    def randn(self, shape) -> Tensor:
        if self.stack is not None:
            out = torch.as_tensor(self.stack[self.i]).float()
            assert tuple(out.shape) == tuple(shape), (out.shape, shape)
        else:
            out = torch.randn(*shape, generator=self.gen)
        self.i += 1
        return out


# ----------------------------------------------------------------------------- single steps
def ddim_step(tb, k: int, x: Tensor, eps_model: Tensor, y: Optional[dict], noise: NoiseSource,
              overlap_len: int, add_blend: bool, clip_denoised: bool = False, tails: Optional[dict] = None, clip_idx: int = 0,
              eta: float = 0.0):
    """One DDIM step at spaced level k incl. the RePaint blend (gaussian_diffusion.py:976-1066; x0 from eps :614-622; eps
    re-derivation :640-644; eta / sigma :1011-1032 — the harness uses eta = 0)."""
    c1, c2 = _f32(tb["sqrt_recip_alphas_cumprod"], k), _f32(tb["sqrt_recipm1_alphas_cumprod"], k)
    ab_prev = _f32(tb["alphas_cumprod_prev"], k)
    x0 = c1 * x - c2 * eps_model
    if clip_denoised:                         # process_xstart (:575-580); the harness always passes False
        x0 = x0.clamp(-1, 1)
    eps = (c1 * x - x0) / c2
    ab = _f32(tb["alphas_cumprod"], k)
    sigma = eta * torch.sqrt((1 - ab_prev) / (1 - ab)) * torch.sqrt(1 - ab / ab_prev)      # (:1011-1015)
    n1 = noise.randn(x.shape)                 # drawn at every step; multiplied by sigma (= 0 at eta = 0) (:1023)
    sample = x0 * torch.sqrt(ab_prev) + torch.sqrt(1 - ab_prev - sigma ** 2) * eps
    if k != 0:                                # nonzero_mask: no noise when (spaced) t == 0 (:1029-1032)
        sample = sample + sigma * n1
    if y and "outpainting_mask" in y and bool(y["outpainting_mask"].any()):
        mask = y["outpainting_mask"]
        nw = torch.sqrt(1 - ab_prev)
        if tails is not None and clip_idx > 0:
            # --same_overlap_noisy (:1040-1042): the previous window's saved noisy tail of this level, no Gaussian draw.  (`tails`
            # is the one dict the reference both reads here and overwrites below: a level revisited by the jump schedule sees
            # what its previous visit in THIS window saved.)
            g = y["gt"].clone()
            g[:, :overlap_len] = tails[k]
        else:
            g = torch.sqrt(ab_prev) * y["gt"] + nw * noise.randn(x.shape)
        if float(nw) < 0.2 and add_blend:
            L = overlap_len
            w = torch.linspace(0, 1, L).view(1, -1, 1)
            g = g.clone()
            g[:, :L] = g[:, :L] * (1 - w) + sample[:, :L] * w
        sample = torch.where(mask, g, sample)
    if tails is not None:
        tails[k] = sample[:, -overlap_len:].clone()          # saved_noisy_tail[str(t)] (:1058-1060)
    return sample, x0


def undo_step(tb, k: int, x: Tensor, noise: NoiseSource) -> Tensor:
    """Re-noise one level with the spaced beta of the level being left (gaussian_diffusion.py:464-473,1274-1278)."""
    beta = _f32(tb["betas"], k)
    return torch.sqrt(1 - beta) * x + torch.sqrt(beta) * noise.randn(x.shape)


def ddpm_step(tb, t: int, x: Tensor, eps_model: Tensor, noise: NoiseSource):
    """Ancestral step, FIXED_SMALL variance (gaussian_diffusion.py:598-600,747-773)."""
    c1, c2 = _f32(tb["sqrt_recip_alphas_cumprod"], t), _f32(tb["sqrt_recipm1_alphas_cumprod"], t)
    x0 = c1 * x - c2 * eps_model
    mean = _f32(tb["posterior_mean_coef1"], t) * x0 + _f32(tb["posterior_mean_coef2"], t) * x
    n = noise.randn(x.shape)
    nz = 0.0 if t == 0 else 1.0
    return mean + nz * torch.exp(0.5 * _f32(tb["posterior_log_variance_clipped"], t)) * n, x0


# ----------------------------------------------------------------------------- loops
EpsFn = Callable[[Tensor, int, Tensor, Tensor], Tensor]


def _call(eps_fn: EpsFn, tb, tmap, k: int, x: Tensor) -> Tensor:
    c1, c2 = _f32(tb["sqrt_recip_alphas_cumprod"], k), _f32(tb["sqrt_recipm1_alphas_cumprod"], k)
    return eps_fn(x, tmap[k], c1, c2)


def ddim_sample_loop(eps_fn: EpsFn, shape, y: Optional[dict], noise: NoiseSource, *, n_steps=1000,
                     spacing="ddim25", jump_length=3, jump_n_sample=5, overlap_len=10,
                     add_blend=True, no_repaint=False, no_resample=False, clip_denoised=False,
                     trace: Optional[list] = None, tails: Optional[dict] = None, clip_idx: int = 0, eta: float = 0.0):
    """ddim_sample_loop dispatch + both progressive loops (gaussian_diffusion.py:1106-1278)."""
    tb, tmap = spaced_tables(n_steps, spacing)
    x = noise.randn(shape)
    masked = bool(y) and "outpainting_mask" in y and bool(y["outpainting_mask"].any())
    if masked and not no_repaint:
        k_resp = int(spacing[4:])
        times = jump_schedule(k_resp) if no_resample else jump_schedule(k_resp, jump_length, jump_n_sample)
        for t_last, t_cur in zip(times[:-1], times[1:]):
            if t_cur < t_last:
                x, x0 = ddim_step(tb, t_last, x, _call(eps_fn, tb, tmap, t_last, x), y, noise, overlap_len, add_blend,
                                  clip_denoised, tails, clip_idx, eta)
                if trace is not None:
                    trace.append(("denoise", t_last, x.clone(), x0.clone()))
            else:
                x = undo_step(tb, t_last, x, noise)
                if trace is not None:
                    trace.append(("undo", t_last, x.clone(), None))
    else:
        for k in range(len(tmap) - 1, -1, -1):
            x, x0 = ddim_step(tb, k, x, _call(eps_fn, tb, tmap, k, x), y, noise, overlap_len, add_blend, clip_denoised, tails, clip_idx, eta)
            if trace is not None:
                trace.append(("denoise", k, x.clone(), x0.clone()))
    return x


def p_sample_loop(eps_fn: EpsFn, shape, noise: NoiseSource, *, n_steps=1000, trace: Optional[list] = None):
    """Plain ancestral loop (gaussian_diffusion.py:923-974); mask-present DDPM is excluded."""
    tb = diffusion_tables(linear_betas(n_steps))
    tmap = list(range(n_steps))
    x = noise.randn(shape)
    for t in range(n_steps - 1, -1, -1):
        x, x0 = ddpm_step(tb, t, x, _call(eps_fn, tb, tmap, t, x), noise)
        if trace is not None:
            trace.append(("ddpm", t, x.clone(), x0.clone()))
    return x


# ----------------------------------------------------------------------------- window chain
def get_windows(x: Tensor, size: int, step: int) -> List[Tensor]:
    """trainers/ddpm_show_trainer.py:801-819 (tensor branch)."""
    n = x.shape[1]
    if n <= size:
        return [x]
    win_num = (n - (size - step)) / float(step)
    out = [x[:, m * step: m * step + size] for m in range(int(win_num))]
    if win_num - int(win_num) != 0:
        out.append(x[:, int(win_num) * step:])
    return out


def window_chain(sample_window: Callable[[int, Tensor, Tensor, dict], Tensor], audio: Tensor, hubert: Tensor,
                 n_poses: int, overlap_len: int, channels: int, fix_very_first_motions: Optional[Tensor] = None) -> Tensor:
    """Sequential out-painting chain (ddpm_show_trainer.py:864-906): window k>0 keeps the last
    ``overlap_len`` frames of window k-1 as its first frames.  ``fix_very_first_motions`` (the clip's ground-truth
    motions) reproduces --fix_very_first (:885-888): window 0 is pinned to ``motions_window0[:, -overlap_len:]``."""
    step = n_poses - overlap_len
    aw, hw = get_windows(audio, n_poses, step), get_windows(hubert, n_poses, step)
    mw = get_windows(fix_very_first_motions, n_poses, step) if fix_very_first_motions is not None else None
    outs, prev = [], None
    for i, (a, h) in enumerate(zip(aw, hw)):
        y = {}
        if overlap_len > 0:
            B, T = a.shape[0], a.shape[1]
            y = {"gt": torch.zeros(B, T, channels), "outpainting_mask": torch.zeros(B, T, channels, dtype=torch.bool)}
            if i == 0 and mw is not None:
                y["outpainting_mask"][:, :overlap_len] = True
                y["gt"][:, :overlap_len] = mw[0][:, -overlap_len:]
            elif i > 0:
                y["outpainting_mask"][:, :overlap_len] = True
                y["gt"][:, :overlap_len] = prev[:, -overlap_len:]
        prev = sample_window(i, a, h, y)
        outs.append(prev if i == len(aw) - 1 else prev[:, :step])
    return torch.cat(outs, dim=1)
