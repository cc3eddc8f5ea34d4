"""Layer-level mirrors of reference modules that the default ``UniDiffuser`` never instantiates.

``LinearTemporalCrossAttention`` (/root/reference/models/transformer.py:133-166) is the ``ca_block`` of the
``model_base='transformer_decoder'`` layer (:294-296, :342-343).  In the reference that model variant cannot be driven end to
end — ``MotionTransformer.forward`` dies in the layer (no ``feat_proj`` is built for the decoder base, :255-289 vs :336; with
classifier-free guidance the null-embedding substitution fails on a shape mismatch first, :330-332) — so the drop-in unit is
the module itself: same constructor arguments, same parameter names, same ``forward(x, xf, emb)``; the work runs in
libdiffsheg_hip.so (``dsh_op_cross_attention``: LayerNorm rows, exact-fp32 MFMA GEMMs, the linear cross-attention core,
LN+FiLM+SiLU rows, residual epilogue).  fp32 only.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from . import _lib

_PARAMS = {"norm_g": "norm.weight", "norm_b": "norm.bias", "text_norm_g": "text_norm.weight", "text_norm_b": "text_norm.bias",
           "wq": "query.weight", "bq": "query.bias", "wk": "key.weight", "bk": "key.bias", "wv": "value.weight", "bv": "value.bias",
           "sty_norm_g": "proj_out.norm.weight", "sty_norm_b": "proj_out.norm.bias",
           "sty_emb_w": "proj_out.emb_layers.1.weight", "sty_emb_b": "proj_out.emb_layers.1.bias",
           "sty_out_w": "proj_out.out_layers.2.weight", "sty_out_b": "proj_out.out_layers.2.bias"}


class LinearTemporalCrossAttention:
    def __init__(self, seq_len: int, latent_dim: int, aud_latent_dim: int, num_head: int, dropout: float, time_embed_dim: int,
                 device="cuda:0"):
        if not torch.cuda.is_available():
            raise _lib.DshError("no GPU visible: diffsheg_amd has no CPU fallback")
        if latent_dim % 64 or aud_latent_dim % 32 or time_embed_dim % 32 or latent_dim % num_head or (latent_dim // num_head) not in (16, 32, 64):
            raise ValueError("latent_dim % 64, aud_latent_dim % 32, time_embed_dim % 32 and head_dim in {16, 32, 64} are required")
        self.latent_dim, self.aud_latent_dim, self.num_head, self.time_embed_dim = latent_dim, aud_latent_dim, num_head, time_embed_dim
        self.device = torch.device(device)
        self._w: Dict[str, torch.Tensor] = {}

    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str = "") -> None:
        D, L, E = self.latent_dim, self.aud_latent_dim, self.time_embed_dim
        shapes = {"norm_g": (D,), "norm_b": (D,), "text_norm_g": (L,), "text_norm_b": (L,), "wq": (D, D), "bq": (D,), "wk": (D, L),
                  "bk": (D,), "wv": (D, L), "bv": (D,), "sty_norm_g": (D,), "sty_norm_b": (D,), "sty_emb_w": (2 * D, E),
                  "sty_emb_b": (2 * D,), "sty_out_w": (D, D), "sty_out_b": (D,)}
        for k, name in _PARAMS.items():
            t = sd[prefix + name]
            if tuple(t.shape) != shapes[k]:
                raise ValueError(f"{prefix + name}: expected {shapes[k]}, got {tuple(t.shape)}")
            self._w[k] = t.detach().to(device=self.device, dtype=torch.float32).contiguous()

    def forward(self, x: torch.Tensor, xf: torch.Tensor, emb: torch.Tensor, mask=None) -> torch.Tensor:
        if not self._w:
            raise _lib.DshError("load_state_dict() first")
        B, T, D = x.shape
        N = xf.shape[1]
        if D != self.latent_dim or xf.shape != (B, N, self.aud_latent_dim) or emb.shape != (B, self.time_embed_dim):
            raise ValueError("x [B,T,latent_dim], xf [B,N,aud_latent_dim], emb [B,time_embed_dim] expected")
        xd, xfd, ed = (t.to(device=self.device, dtype=torch.float32).contiguous() for t in (x, xf, emb))
        y = torch.empty_like(xd)
        w = _lib.CrossAttnWeightsC(**{k: v.data_ptr() for k, v in self._w.items()})
        st = torch.cuda.current_stream(self.device)
        _lib.check(_lib.lib().dsh_op_cross_attention(C.c_void_p(st.cuda_stream), C.byref(w), xd.data_ptr(), xfd.data_ptr(), ed.data_ptr(),
                                                     B, T, N, D, self.aud_latent_dim, self.time_embed_dim, self.num_head, y.data_ptr()),
                   "dsh_op_cross_attention")
        return y

    __call__ = forward
