"""Host-side mirror of the reference denoiser interface (boundary 1).

``UniDiffuser`` here is *not* an nn.Module: it is a thin handle on a ``dsh_ctx`` of
libdiffsheg_hip.so that keeps the reference's call signature

    model(x, timesteps, sqrt_alphas, audio_emb, length, person_id, add_cond={}, pe_type=..., y=None)

(/root/reference/models/transformer.py:728) so ``GaussianDiffusion.p_mean_variance`` style callers
(``model(x, ts, **model_kwargs)``, gaussian_diffusion.py:536) work unchanged.  PyTorch is used only
for device memory and the stream handle.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterator, Optional

import torch

from . import _lib
from .config import DiffSHEGConfig
from .weights import strip_ddp_prefix, validate_state_dict

_PRECISION = {"fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1}


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _dev_f32(t: torch.Tensor, device: torch.device) -> torch.Tensor:
    return t.to(device=device, dtype=torch.float32).contiguous()


class UniDiffuser:
    """MI355X UniDiffuser: ``encoder_aud`` + ``encoder_exp`` + ``encoder_ges`` behind one C handle."""
    _UNIDIFFUSER = True

    def __init__(self, cfg: DiffSHEGConfig, state_dict: Dict[str, torch.Tensor], device="cuda:0",
                 precision: str = "fp32"):
        if not torch.cuda.is_available():
            raise _lib.DshError("no GPU visible: diffsheg_amd has no CPU fallback (the CPU oracle lives in oracle/ "
                                "and is test infrastructure only)")
        if precision not in _PRECISION:
            raise ValueError(f"precision must be one of {sorted(_PRECISION)}")
        self.cfg = cfg
        self.device = torch.device(device)
        self.precision = precision
        self._lib = _lib.lib()
        torch.cuda.set_device(self.device)
        self._stream = torch.cuda.current_stream(self.device)
        mc = _lib.ModelConfigC(cfg.dim_pose, cfg.expression_dim, cfg.style_dim, int(cfg.classifier_free),
                               float(cfg.cond_scale), cfg.latent_dim, cfg.ff_size, cfg.num_layers, cfg.num_heads,
                               cfg.audio_dim, cfg.aud_latent_dim, cfg.hubert_dim, cfg.hubert_enc_dim,
                               _PRECISION[precision], int(not cfg.unidiffuser))
        if cfg.unidiffuser != self._UNIDIFFUSER:
            raise ValueError(f"{type(self).__name__} needs a config with unidiffuser={self._UNIDIFFUSER} (runner.py:33-57 picks the class by opt.unidiffuser)")
        h = C.c_void_p()
        _lib.check(self._lib.dsh_create(C.byref(mc), C.c_void_p(self._stream.cuda_stream), C.byref(h)), "dsh_create")
        self._h = h
        self._cond_key = None
        self._cond_keep = None
        self._dummy = torch.zeros(1, device=self.device)
        self.load_state_dict(state_dict)

    # ---- weights ---------------------------------------------------------------------------
    def load_state_dict(self, state_dict: Dict[str, torch.Tensor]) -> None:
        sd = strip_ddp_prefix(state_dict)
        validate_state_dict(self.cfg, sd)
        for name, t in sd.items():
            if not torch.is_floating_point(t):
                continue                      # num_batches_tracked
            a = t.detach().to("cpu", torch.float32).contiguous()
            shape = (C.c_int64 * a.dim())(*a.shape)
            _lib.check(self._lib.dsh_load_tensor(self._h, name.encode(), C.c_void_p(a.data_ptr()), shape, a.dim()),
                       f"dsh_load_tensor({name})")
        _lib.check(self._lib.dsh_finalize_weights(self._h), "dsh_finalize_weights")

    @property
    def weight_bytes(self) -> int:
        return int(self._lib.dsh_weight_bytes(self._h))

    # ---- nn.Module look-alikes used by the reference sampler/harness ---------------------------
    def parameters(self) -> Iterator[torch.Tensor]:
        yield self._dummy                     # `next(model.parameters()).device` (gaussian_diffusion.py:1181)

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dsh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- stream ordering ---------------------------------------------------------------------
    def _enter(self) -> "torch.cuda.Stream":
        """The context is bound to the stream that was current at construction.  When the caller now works under
        another torch stream, order the context stream after it (and, in :meth:`_exit`, the caller's stream after the
        context's work), so inputs are complete before the native kernels read them and outputs / freed temporaries
        are not touched early.  A context built on the NULL stream runs on a private *blocking* stream, which the NULL
        stream orders implicitly, so waiting on / for ``torch.cuda.default_stream()`` is sufficient there as well."""
        cur = torch.cuda.current_stream(self.device)
        if cur != self._stream:
            self._stream.wait_stream(cur)
        return cur

    def _exit(self, cur: "torch.cuda.Stream") -> None:
        if cur != self._stream:
            cur.wait_stream(self._stream)

    # ---- conditioning ------------------------------------------------------------------------
    def set_condition(self, audio_emb: torch.Tensor, person_id: torch.Tensor, hubert: torch.Tensor) -> None:
        """Upload the step-invariant conditioning and run hubert_encoder / pid_embed once."""
        B, T = int(audio_emb.shape[0]), int(audio_emb.shape[1])
        if person_id.dim() == 1:                                   # transformer.py:502-503
            person_id = person_id.unsqueeze(0)
        if person_id.shape[0] != B:
            if person_id.shape[0] != 1:
                raise ValueError("person_id batch does not match audio_emb")
            person_id = person_id.expand(B, -1)
        if tuple(audio_emb.shape) != (B, T, self.cfg.audio_dim):
            raise ValueError(f"audio_emb must be [B,T,{self.cfg.audio_dim}], got {tuple(audio_emb.shape)}")
        if tuple(hubert.shape) != (B, T, self.cfg.hubert_dim):
            raise ValueError(f"pretrain_aud_feat must be [B,T,{self.cfg.hubert_dim}], got {tuple(hubert.shape)}")
        if person_id.shape[1] != self.cfg.style_dim:
            raise ValueError(f"person_id must be [B,{self.cfg.style_dim}]")
        a, p, hb = (_dev_f32(t, self.device) for t in (audio_emb, person_id, hubert))
        cur = self._enter()
        _lib.check(self._lib.dsh_set_condition(self._h, B, T, a.data_ptr(), p.data_ptr(), hb.data_ptr()),
                   "dsh_set_condition")
        self._exit(cur)
        self._cond_keep = (a, p, hb)           # (the library copies them in stream order; kept for the allocator's sake)
        self.batch, self.frames = B, T

    def _maybe_set_condition(self, audio_emb, person_id, add_cond) -> None:
        if "pretrain_aud_feat" not in (add_cond or {}):
            raise ValueError("add_cond['pretrain_aud_feat'] (HuBERT features) is required (addHubert=True)")
        hub = add_cond["pretrain_aud_feat"]
        # Identity + in-place-version check on the caller's tensor objects.  The objects themselves are
        # kept alive in the key: a freed tensor's address can be handed to a new tensor of the same shape,
        # so data_ptr alone would alias stale conditioning.
        src = (audio_emb, person_id, hub)
        vers = tuple(t._version for t in src)
        if (self._cond_key is None or any(a is not b for a, b in zip(self._cond_key[0], src))
                or self._cond_key[1] != vers):
            self.set_condition(audio_emb, person_id, hub)
            self._cond_key = (src, vers)

    # ---- boundary 1 -----------------------------------------------------------------------------
    def __call__(self, x, timesteps, sqrt_alphas=None, audio_emb=None, length=None, person_id=None, add_cond=None,
                 pe_type="pe_sinu", y=None) -> torch.Tensor:
        return self.forward(x, timesteps, sqrt_alphas, audio_emb, length, person_id, add_cond, pe_type, y)

    def forward(self, x, timesteps, sqrt_alphas, audio_emb, length, person_id, add_cond=None, pe_type="pe_sinu",
                y=None) -> torch.Tensor:
        if pe_type not in ("pe_sinu",):
            raise NotImplementedError(f"pe_type={pe_type!r}: only the default 'pe_sinu' path is built")
        if sqrt_alphas is None or len(sqrt_alphas) != 2:
            raise ValueError("sqrt_alphas=[sqrt_recip_alphas_cumprod_t, sqrt_recipm1_alphas_cumprod_t] is required")
        self._maybe_set_condition(audio_emb, person_id, add_cond)
        B, T, Cc = x.shape
        if (B, T) != (self.batch, self.frames) or Cc != self.cfg.net_dim_pose:
            raise ValueError(f"x shape {tuple(x.shape)} does not match conditioning ({self.batch},{self.frames},{self.cfg.net_dim_pose})")
        xd = _dev_f32(x, self.device)
        td = timesteps.to(device=self.device, dtype=torch.int64).contiguous()
        c1 = _dev_f32(sqrt_alphas[0].reshape(B, -1)[:, 0], self.device)
        c2 = _dev_f32(sqrt_alphas[1].reshape(B, -1)[:, 0], self.device)
        out = torch.empty_like(xd)
        cur = self._enter()
        _lib.check(self._lib.dsh_eval(self._h, xd.data_ptr(), td.data_ptr(), c1.data_ptr(), c2.data_ptr(),
                                      out.data_ptr()), "dsh_eval")
        self._exit(cur)
        return out

    # ---- introspection -----------------------------------------------------------------------------
    def eval_flops(self) -> float:
        return float(self._lib.dsh_eval_flops(self._h))

    def debug_tap(self, what: str) -> torch.Tensor:
        w = {"aud_feat": self.cfg.audio_dim, "expr_x0": self.cfg.expression_dim}[what]
        out = torch.empty(self.batch, self.frames, w, device=self.device)
        _lib.check(self._lib.dsh_debug_copy(self._h, what.encode(), out.data_ptr()), "dsh_debug_copy")
        return out


class MotionTransformer(UniDiffuser):
    """The model ``runner.py:46-57`` builds when ``opt.unidiffuser`` is False (``model_base='transformer_encoder'``): ONE
    motion transformer over all ``net_dim_pose`` channels.  Same native context (``single_transformer = 1``), state-dict keys
    without the ``encoder_*`` prefix, and the reference's call signature (transformer.py:496)

        model(x, timesteps, audio_emb, length, person_id, add_cond={}, pe_type=..., y=None, block=None)

    — no ``sqrt_alphas`` (gaussian_diffusion.py:527-536 only passes them for the UniDiffuser)."""
    _UNIDIFFUSER = False

    def __call__(self, x, timesteps, audio_emb=None, length=None, person_id=None, add_cond=None, pe_type="pe_sinu", y=None,
                 block=None, sqrt_alphas=None) -> torch.Tensor:
        return self.forward(x, timesteps, audio_emb, length, person_id, add_cond, pe_type, y, block)

    def forward(self, x, timesteps, audio_emb, length, person_id, add_cond=None, pe_type="pe_sinu", y=None, block=None) -> torch.Tensor:
        one = torch.ones(x.shape[0], device=self.device)
        return UniDiffuser.forward(self, x, timesteps, [one, one], audio_emb, length, person_id, add_cond, pe_type, y)

    def debug_tap(self, what: str) -> torch.Tensor:
        raise NotImplementedError("aud_feat / expr_x0 taps exist only in the UniDiffuser model")
