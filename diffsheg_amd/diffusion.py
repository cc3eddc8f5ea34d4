"""Host-side mirror of the reference sampler interface (boundary 2).

Same class / function names and call signatures as /root/reference/models/gaussian_diffusion.py
(``GaussianDiffusion.p_sample_loop`` :776, ``ddim_sample_loop`` :1106) and models/respace.py
(``space_timesteps`` :7, ``SpacedDiffusion`` :60); the loops themselves run natively in
libdiffsheg_hip.so (csrc/sampler.hip) with no host syncs per step.

Differences a caller can observe, all loud:
  * the model must be a :class:`diffsheg_amd.model.UniDiffuser` (epsilon prediction, FIXED_SMALL var);
  * ``denoised_fn`` / ``cond_fn`` / ``pre_seq`` / ``transl_req`` and an ``opt.cond_scale`` other than the model
    handle's raise NotImplementedError (``same_overlap_noisy``, ``eta != 0`` and ``fix_head_var`` are built: the
    saved noisy tails live in the native context, eta adds one draw per DDIM step, fix_head_var is a no-op of
    the reference's own sampling code — see ``_run``);
  * Gaussian noise comes from ``noise_source`` (any object with ``randn(shape) -> Tensor``, consumed in
    the reference's draw order — this is how parity tests inject identical noise) or, if None, from the
    on-device Philox generator seeded by ``seed`` / ``torch.initial_seed()``; ``row_keys`` (one integer per
    batch row) additionally gives every row its own Philox stream, so that a chain draws the same noise in
    whatever batch / on whatever rank it is sampled (the sharded long-audio path).
"""
from __future__ import annotations

import ctypes as C
import enum
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from .model import UniDiffuser

_TABLES = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
           "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
           "posterior_mean_coef1", "posterior_mean_coef2"]


class ModelMeanType(enum.Enum):
    PREVIOUS_X = enum.auto()
    START_X = enum.auto()
    EPSILON = enum.auto()


class ModelVarType(enum.Enum):
    LEARNED = enum.auto()
    FIXED_SMALL = enum.auto()
    FIXED_LARGE = enum.auto()
    LEARNED_RANGE = enum.auto()


def get_named_beta_schedule(schedule_name: str, num_diffusion_timesteps: int) -> np.ndarray:
    """gaussian_diffusion.py:234-251 — only 'linear' is on the path."""
    if schedule_name != "linear":
        raise NotImplementedError(f"unknown beta schedule: {schedule_name}")
    return diffusion_table(num_diffusion_timesteps, 0, "betas")


def diffusion_table(steps: int, respacing: int, name: str) -> np.ndarray:
    buf = (C.c_double * max(steps, 1))()
    n = _lib.check(_lib.lib().dsh_diffusion_table(steps, respacing, name.encode(), buf, steps), "dsh_diffusion_table")
    return np.array(buf[:n], dtype=np.float64)


def space_timesteps(num_timesteps: int, section_counts) -> set:
    """respace.py:7-57 — the hot path only ever uses the 'ddimN' form."""
    if isinstance(section_counts, str) and section_counts.startswith("ddim"):
        k = int(section_counts[len("ddim"):])
        buf = (C.c_int32 * num_timesteps)()
        n = _lib.check(_lib.lib().dsh_timestep_map(num_timesteps, k, buf, num_timesteps), "dsh_timestep_map")
        return set(buf[:n])
    raise NotImplementedError("only 'ddimN' respacing is built (trainers hard-code 'ddim25')")


def get_schedule_jump_cjm_ddim(time_respacing: int = 25, jump_length: int = 1, jump_n_sample: int = 1) -> List[int]:
    """scheduler.py:178-208."""
    buf = (C.c_int32 * 4096)()
    n = _lib.check(_lib.lib().dsh_jump_schedule(time_respacing, jump_length, jump_n_sample, buf, 4096), "dsh_jump_schedule")
    return list(buf[:n])


class _SavedTails:
    """Opaque stand-in for the reference's ``saved_noisy_tail`` dict: the tensors stay in the native context."""

    def __init__(self, model):
        self.model = model


class GaussianDiffusion:
    """Full-chain sampler handle (ancestral DDPM): mirrors gaussian_diffusion.py:300-390."""

    _respacing = 0

    def __init__(self, *, opt, betas=None, model_mean_type=ModelMeanType.EPSILON,
                 model_var_type=ModelVarType.FIXED_SMALL, loss_type=None, rescale_timesteps=False,
                 num_timesteps: Optional[int] = None):
        if model_mean_type not in (ModelMeanType.EPSILON, "epsilon"):
            raise NotImplementedError("only epsilon prediction is built (trainers use model_mean_type='epsilon')")
        if model_var_type not in (ModelVarType.FIXED_SMALL,):
            raise NotImplementedError("only FIXED_SMALL variance is built")
        if rescale_timesteps:
            raise NotImplementedError("rescale_timesteps=True is not on the path")
        self.opt = opt
        self.model_mean_type, self.model_var_type, self.loss_type = model_mean_type, model_var_type, loss_type
        self.rescale_timesteps = False
        n = int(num_timesteps if num_timesteps is not None else (len(betas) if betas is not None else 1000))
        if betas is not None:
            ref = diffusion_table(n, 0, "betas")
            if not np.allclose(np.asarray(betas, dtype=np.float64), ref, rtol=0, atol=1e-15):
                raise NotImplementedError("only the 'linear' beta schedule is built")
        self.original_num_steps = n
        self._load_tables()

    def _load_tables(self):
        for name in _TABLES:
            setattr(self, name, diffusion_table(self.original_num_steps, self._respacing, name))
        self.num_timesteps = len(self.betas)
        self.timestep_map = list(range(self.num_timesteps))

    # ---- helpers --------------------------------------------------------------------------
    def _opts(self, kind: int, clip_denoised: bool, noise_mode: int, seed: int, clip_idx: int = 0, eta: float = 0.0) -> _lib.SamplerOptsC:
        o = self.opt
        return _lib.SamplerOptsC(kind, self.original_num_steps, max(self._respacing, 1),
                                 int(getattr(o, "jump_length", 3)), int(getattr(o, "jump_n_sample", 5)),
                                 int(getattr(o, "overlap_len", 0)), int(bool(getattr(o, "addBlend", getattr(o, "add_blend", True)))),
                                 int(bool(getattr(o, "no_resample", False))), int(bool(getattr(o, "no_repaint", False))),
                                 int(bool(clip_denoised)), noise_mode, seed & 0xFFFFFFFFFFFFFFFF,
                                 int(bool(getattr(o, "same_overlap_noisy", False))), int(clip_idx), float(eta))

    def _run(self, kind, model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs, eta=0.0,
             noise_source=None, seed=None, return_trace=False, row_keys=None):
        if not isinstance(model, UniDiffuser):
            raise TypeError("model must be a diffsheg_amd.model.UniDiffuser (no generic-callable / CPU fallback)")
        if denoised_fn is not None or cond_fn is not None:
            raise NotImplementedError("denoised_fn / cond_fn are not on the accelerated path")
        if eta != 0.0 and kind != 0:
            raise TypeError("eta is an argument of the DDIM loops only")
        if model_kwargs is None or model_kwargs.get("y", None) is None:
            # the reference dereferences model_kwargs['y'].keys() (gaussian_diffusion.py:810,1126)
            raise AttributeError("'NoneType' object has no attribute 'keys' (model_kwargs['y'] must be a dict)")
        y = model_kwargs["y"]
        # options of the reference's `opt` namespace that change results and are not built: refuse, never ignore
        # opt.fix_head_var (gaussian_diffusion.py:444,759): q_sample (training / pre_seq only) zeroes the noise of the head channels;
        # in p_sample the mask it edits is [B, 1, 1], so `nonzero_mask[..., 90:] = 0` selects nothing, and ddim_sample never reads
        # the switch — sampling is bit-identical with it on (checked against the imported reference: tests/golden/make_golden.py
        # gen_eta_fhv, fixture ddpm50_fhv_show.npz).  What remains of it on this path is the reference's own refusal of other datasets.
        if getattr(self.opt, "fix_head_var", False) and kind == 1 and getattr(self.opt, "dataset_name", None) not in ("freeform_all", "talkshow"):
            raise NotImplementedError("fix_head_var: dataset_name must be 'freeform_all' or 'talkshow' (gaussian_diffusion.py:760-766)")
        son = bool(getattr(self.opt, "same_overlap_noisy", False))
        if son and kind != 0:
            raise NotImplementedError("same_overlap_noisy only exists in the DDIM loop (gaussian_diffusion.py:1040-1060)")
        clip_idx = int(y.get("clip_idx", 0)) if son else 0
        cs = getattr(self.opt, "cond_scale", None)
        if cs is not None and float(cs) != float(model.cfg.cond_scale):
            raise NotImplementedError(f"opt.cond_scale={cs} differs from the scale baked into the model handle "
                                      f"({model.cfg.cond_scale}): build the UniDiffuser with get_config(..., cond_scale={cs})")
        B, T, Cc = (int(s) for s in shape)
        dev = model.device
        model._maybe_set_condition(model_kwargs["audio_emb"], model_kwargs["person_id"], model_kwargs.get("add_cond"))
        if (B, T) != (model.batch, model.frames) or Cc != model.cfg.net_dim_pose:
            raise ValueError(f"shape {tuple(shape)} does not match the conditioning")
        gt = mask = None
        masked = False
        if "outpainting_mask" in y:
            mask = y["outpainting_mask"].to(device=dev)
            # the reference's `True in mask` costs one host sync per window; a harness that built the mask itself says what
            # it holds (key "outpainting_mask_any", set by trainer.sample_arbitrary_len) and no sync is needed
            hint = y.get("outpainting_mask_any", None)
            masked = bool(hint) if hint is not None else bool(mask.any().item())
            if masked:
                if kind == 1:
                    raise NotImplementedError("mask-present DDPM sampling (hard-wired 250-step schedule) is excluded")
                # the kernels index mask / gt as dense [B,T,C]: broadcast what the reference would broadcast, reject the rest
                try:
                    mask = mask.expand(B, T, Cc).to(torch.uint8).contiguous()
                    gt = y["gt"].to(device=dev, dtype=torch.float32).expand(B, T, Cc).contiguous()
                except RuntimeError as e:
                    raise ValueError(f"outpainting_mask {tuple(y['outpainting_mask'].shape)} / gt {tuple(y['gt'].shape)} "
                                     f"do not broadcast to the sample shape {(B, T, Cc)}") from e
        init = noise is not None
        x = noise.to(device=dev, dtype=torch.float32).contiguous().clone() if init else torch.empty(B, T, Cc, device=dev)
        mode = 0 if noise_source is not None else 1
        if seed is None:
            seed = int(torch.initial_seed()) + GaussianDiffusion._calls
            GaussianDiffusion._calls += 1
        opts = self._opts(kind, clip_denoised, mode, int(seed), clip_idx, eta)
        lib = _lib.lib()
        n_draws = _lib.check(lib.dsh_sample_num_draws(C.byref(opts), int(masked), int(init)), "dsh_sample_num_draws")
        stack = None
        if noise_source is not None:
            stack = torch.empty(n_draws, B * T * Cc, device=dev)
            for i in range(n_draws):
                stack[i].copy_(noise_source.randn((B, T, Cc)).reshape(-1), non_blocking=False)
        trace = None
        if return_trace:
            n_steps = _lib.check(lib.dsh_sample_num_steps(C.byref(opts), int(masked)), "dsh_sample_num_steps")
            trace = torch.empty(n_steps, B, T, Cc, device=dev)
        # Philox noise: optional per-row generator keys (a chain's global id), see dsh_sample_set_row_keys
        if row_keys is not None and len(row_keys) != B:
            raise ValueError(f"row_keys needs one key per batch row ({B}), got {len(row_keys)}")
        nk = 0 if (row_keys is None or noise_source is not None) else B
        karr = (C.c_uint64 * max(nk, 1))(*([int(k) & 0xFFFFFFFFFFFFFFFF for k in row_keys] if nk else [0]))
        cur = model._enter()
        try:
            _lib.check(lib.dsh_sample_set_row_keys(model._h, karr, nk), "dsh_sample_set_row_keys")
            try:
                _lib.check(lib.dsh_sample(model._h, C.byref(opts), x.data_ptr(), int(init),
                                          None if gt is None else gt.data_ptr(), None if not masked else mask.data_ptr(),
                                          int(masked), None if stack is None else stack.data_ptr(), n_draws,
                                          None if trace is None else trace.data_ptr()), "dsh_sample")
            except Exception:
                if nk:                                   # the keys are sticky in the native context: do not leak them into the next call
                    lib.dsh_sample_set_row_keys(model._h, (C.c_uint64 * 1)(0), 0)
                raise
        finally:
            model._exit(cur)
        self._keep = (gt, mask, stack)          # consumed asynchronously on the stream
        if son:
            # gaussian_diffusion.py:1155-1157: the loop returns the dict of the last step plus the saved tails.  The tails live
            # in the native context (one slot per spaced level, overwritten step by step exactly like the reference's dict,
            # which IS the object the next window receives as y['previous_noisy_tail']); the handle only documents that.
            out = {"sample": x, "saved_noisy_tail": _SavedTails(model)}
            return (out, trace) if return_trace else out
        return (x, trace) if return_trace else x

    _calls = 0

    # ---- boundary 2 ------------------------------------------------------------------------
    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                      model_kwargs=None, device=None, pre_seq=None, transl_req=None, progress=False, **kw):
        """gaussian_diffusion.py:776-841 (plain ancestral loop over all timesteps)."""
        if pre_seq is not None or transl_req is not None:
            raise NotImplementedError("pre_seq / transl_req are not on the accelerated path")
        if self._respacing:
            raise NotImplementedError("p_sample_loop on a SpacedDiffusion is not used by the harness")
        return self._run(1, model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs, **kw)

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                         model_kwargs=None, device=None, progress=False, eta=0.0, **kw):
        """gaussian_diffusion.py:1106-1159 (dispatches to the harmonize schedule when a mask is set)."""
        if not self._respacing:
            raise NotImplementedError("ddim_sample_loop needs a SpacedDiffusion('ddimK') (trainers use 'ddim25')")
        return self._run(0, model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs, eta=eta, **kw)


class SpacedDiffusion(GaussianDiffusion):
    """respace.py:60-110: tables rebuilt from the kept cumulative alphas; the model sees ORIGINAL timesteps."""

    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        n = len(kwargs["betas"]) if kwargs.get("betas") is not None else int(kwargs.get("num_timesteps", 1000))
        k = len(self.use_timesteps)
        if self.use_timesteps != space_timesteps(n, f"ddim{k}"):
            raise NotImplementedError("only 'ddimK' strided subsets are built")
        self._respacing = k
        super().__init__(**kwargs)

    def _load_tables(self):
        super()._load_tables()
        self.timestep_map = sorted(self.use_timesteps)
