"""Static configuration of the DiffSHEG sampling hot path.

The reference reads ~40 attributes off a shared, *mutable* argparse namespace that it
toggles in the middle of a forward pass (/root/reference/models/transformer.py:743-748,
764-765).  Here the two dataset configurations are baked into an immutable struct; the
values come from the dataset constants in /root/reference/runner.py:124-222 and the
defaults in /root/reference/options/base_options.py:13-148.
"""
from __future__ import annotations

from dataclasses import dataclass, replace


@dataclass(frozen=True)
class DiffSHEGConfig:
    dataset: str            # "show" | "beat"
    dim_pose: int           # gesture channels  (runner.py:203 / :137)
    expression_dim: int     # expression channels
    style_dim: int          # one-hot speaker width
    n_poses: int            # frames per window
    classifier_free: bool
    cond_scale: float
    # architecture constants (runner.py:34-45, transformer.py:349-369)
    audio_dim: int = 128        # mel feature width == encoder_aud latent
    latent_dim: int = 512
    ff_size: int = 1024
    num_layers: int = 8
    num_heads: int = 8
    aud_latent_dim: int = 256   # audio_proj output width
    hubert_dim: int = 1024
    hubert_enc_dim: int = 128
    pe_period: int = 600
    pe_max_len: int = 1200
    # sampler (options/base_options.py:118-128, trainers/ddpm_show_trainer.py:73)
    diffusion_steps: int = 1000
    timestep_respacing: str = "ddim25"
    overlap_len: int = 10
    jump_length: int = 3
    jump_n_sample: int = 5
    add_blend: bool = True
    no_resample: bool = False
    no_repaint: bool = False
    # options/base_options.py:101 --unidiffuser (default True).  False = ONE MotionTransformer over all net_dim_pose channels
    # (runner.py:46-57, model_base 'transformer_encoder'): no encoder_aud, audio_proj on the 128 mel features
    unidiffuser: bool = True

    # ---- derived ----
    @property
    def split_pos(self) -> int:
        return self.dim_pose

    @property
    def net_dim_pose(self) -> int:          # C = gesture | expression
        return self.dim_pose + self.expression_dim

    @property
    def time_embed_dim(self) -> int:
        return 4 * self.latent_dim

    @property
    def cfg_active(self) -> bool:
        """CFG batch doubling happens only in eval with cond_scale != 1 (transformer.py:537)."""
        return self.classifier_free and self.cond_scale != 1

    @property
    def concat_dim_exp(self) -> int:        # h | audio_proj | hubert128          (transformer.py:399-419)
        return self.latent_dim + self.aud_latent_dim + self.hubert_enc_dim

    @property
    def concat_dim_single(self) -> int:     # h | audio_proj | hubert128 (single MotionTransformer: same widths as encoder_exp)
        return self.concat_dim_exp

    @property
    def concat_dim_ges(self) -> int:        # h | audio_proj | hubert128 | expr_x0
        return self.concat_dim_exp + self.expression_dim

    def with_(self, **kw) -> "DiffSHEGConfig":
        return replace(self, **kw)


def show_config(**kw) -> DiffSHEGConfig:
    """SHOW / talkshow: C=232 (129|103), T=88, S=4, CFG 1.25 (runner.py:189-222, README.md:120-137)."""
    base = DiffSHEGConfig(dataset="show", dim_pose=129, expression_dim=103, style_dim=4, n_poses=88,
                          classifier_free=True, cond_scale=1.25, overlap_len=10)
    return replace(base, **kw)


def beat_config(**kw) -> DiffSHEGConfig:
    """BEAT: C=192 (141|51), T=34, S=30, no CFG (runner.py:124-169, train_test_scripts.sh:19-31)."""
    base = DiffSHEGConfig(dataset="beat", dim_pose=141, expression_dim=51, style_dim=30, n_poses=34,
                          classifier_free=False, cond_scale=1.0, overlap_len=4)
    return replace(base, **kw)


def get_config(name: str, **kw) -> DiffSHEGConfig:
    if name == "show":
        return show_config(**kw)
    if name == "beat":
        return beat_config(**kw)
    raise ValueError(f"unknown dataset config {name!r}")
