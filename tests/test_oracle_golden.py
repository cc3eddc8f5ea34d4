"""CPU tests (-m "not gpu"): the oracle restatement vs the golden vectors produced by the imported
reference (tests/golden/make_golden.py).  This is what pins the oracle: the reference has no tests
or golden vectors of its own."""
import json
import os

import numpy as np
import pytest
import torch

from diffsheg_amd.config import get_config
from diffsheg_amd.synthetic import make_inputs
from oracle import denoiser_ref as D
from oracle import sampler_ref as S
from util import GOLDEN, golden, synthetic_sd

torch.set_num_threads(min(8, os.cpu_count() or 1))


# ---- S1/S2/S3: tables, respacing, schedules ------------------------------------------------------
def test_tables_match_reference():
    full = golden("tables_ddpm1000.npz")
    tb = S.diffusion_tables(S.linear_betas(1000))
    for k in full.files:
        np.testing.assert_allclose(tb[k], full[k], rtol=0, atol=0, err_msg=k)
    sp = golden("tables_ddim25.npz")
    tbs, tmap = S.spaced_tables(1000, "ddim25")
    assert tmap == list(sp["timestep_map"]) == list(range(0, 1000, 40))
    for k in sp.files:
        if k != "timestep_map":
            np.testing.assert_allclose(tbs[k], sp[k], rtol=0, atol=0, err_msg=k)
    # anchors from SURVEY appendix A.1
    assert abs(tbs["alphas_cumprod"][14] - 0.040879) < 1e-6 and abs(tbs["betas"][14] - 0.35406) < 1e-5


def test_jump_schedules_match_reference():
    ref = json.load(open(os.path.join(GOLDEN, "schedules.json")))
    for key, want in ref.items():
        if key.startswith("resp20"):
            got = S.jump_schedule(20, 3, 5)
        else:
            jl, jn = (int(v) for v in key.split(","))
            got = S.jump_schedule(25, jl, jn)
        assert got == want, key
    t = S.jump_schedule(25, 3, 5)
    pairs = list(zip(t[:-1], t[1:]))
    assert sum(b < a for a, b in pairs) == 63 and sum(b > a for a, b in pairs) == 48


# ---- D1-D7: denoiser ------------------------------------------------------------------------------
def test_per_op_fixtures():
    f = golden("ops_show.npz")
    cfg = get_config("show")
    sd = synthetic_sd("show")
    B, T = int(f["B"]), int(f["T"])
    g = torch.Generator().manual_seed(int(f["seed"]))
    emb = torch.randn(2 * B, cfg.time_embed_dim, generator=g) * 0.5
    h = torch.randn(2 * B, T, cfg.latent_dim, generator=g)
    p = "encoder_exp.temporal_decoder_blocks.3"
    a = torch.randn(2 * B, T, cfg.aud_latent_dim, generator=g)
    hub = torch.randn(2 * B, T, cfg.hubert_enc_dim, generator=g)
    with torch.no_grad():
        assert torch.equal(D.timestep_embedding(torch.tensor([0, 40, 560, 999]), 512), torch.from_numpy(f["temb"]))
        np.testing.assert_allclose(D.stylization(sd, p + ".sa_block.proj_out", h, emb), f["stylization"], atol=1e-6)
        np.testing.assert_allclose(D.linear_self_attention(sd, p + ".sa_block", h, emb, 8), f["self_attn"], atol=1e-6)
        np.testing.assert_allclose(D.ffn(sd, p + ".ffn", h, emb), f["ffn"], atol=1e-6)
        cond = torch.cat((a, hub), -1)
        null = sd["encoder_exp.null_cond_emb"]
        np.testing.assert_allclose(D.decoder_layer(sd, p, h, cond, emb, 8, null, True), f["layer_cfg"], atol=1e-5)
        np.testing.assert_allclose(D.decoder_layer(sd, p, h, cond, emb, 8, null, False), f["layer_nocfg"], atol=1e-5)
        hubert = torch.randn(B, T, cfg.hubert_dim, generator=g)
        np.testing.assert_allclose(D.hubert_encoder(sd, "encoder_ges.hubert_encoder", hubert), f["hubert_enc"], atol=1e-5)
        aud = torch.randn(B, T, cfg.audio_dim, generator=g)
        np.testing.assert_allclose(D.decoder_layer(sd, "encoder_aud", aud, None, emb[:B], 8, None, False),
                                   f["encoder_aud"], atol=1e-5)


def test_per_op_fixtures_at_the_production_window_length():
    """ops_show_t88.npz: the reference's self-attention block, FFN block and decoder layer (CFG halves) at T = 88, B' = 2"""
    f = golden("ops_show_t88.npz")
    cfg = get_config("show")
    sd = synthetic_sd("show")
    B, T = int(f["B"]), int(f["T"])
    g = torch.Generator().manual_seed(int(f["seed"]))
    emb = torch.randn(2 * B, cfg.time_embed_dim, generator=g) * 0.5
    h = torch.randn(2 * B, T, cfg.latent_dim, generator=g)
    p = "encoder_exp.temporal_decoder_blocks.3"
    a = torch.randn(2 * B, T, cfg.aud_latent_dim, generator=g)
    hub = torch.randn(2 * B, T, cfg.hubert_enc_dim, generator=g)
    with torch.no_grad():
        np.testing.assert_allclose(D.linear_self_attention(sd, p + ".sa_block", h, emb, 8), f["self_attn"], atol=1e-6)
        np.testing.assert_allclose(D.ffn(sd, p + ".ffn", h, emb), f["ffn"], atol=1e-6)
        np.testing.assert_allclose(D.decoder_layer(sd, p, h, torch.cat((a, hub), -1), emb, 8, sd["encoder_exp.null_cond_emb"], True), f["layer_cfg"], atol=1e-5)


@pytest.mark.parametrize("ds", ["beat", "show"])
def test_full_eval_matches_reference(ds):
    cfg = get_config(ds)
    sd = synthetic_sd(ds)
    f = golden(f"eval_{ds}.npz")
    B = int(f["batch"])
    inp = make_inputs(cfg, B, seed=int(f["input_seed"]))
    for tag in ["k0", "k14", "t999"]:
        t = torch.full((B,), int(f[f"{tag}_t"]))
        c1, c2 = torch.tensor(np.float32(f[f"{tag}_c1"])), torch.tensor(np.float32(f[f"{tag}_c2"]))
        with torch.no_grad():
            eps, parts = D.unidiffuser(sd, cfg, inp["x_T"], t, c1, c2, inp["audio_emb"], inp["person_id"],
                                       inp["pretrain_aud_feat"], return_parts=True)
        np.testing.assert_allclose(eps, f[f"{tag}_eps"], atol=2e-6)
        np.testing.assert_allclose(parts["eps_exp"], f[f"{tag}_eps_exp"], atol=2e-6)


# ---- S4-S8, H1/H2: sampling loops ---------------------------------------------------------------------
def _eps_fn(ds, inp):
    cfg, sd = get_config(ds), synthetic_sd(ds)
    B = inp["audio_emb"].shape[0]

    def fn(x, t, c1, c2):
        with torch.no_grad():
            return D.unidiffuser(sd, cfg, x, torch.full((B,), t), c1, c2, inp["audio_emb"], inp["person_id"],
                                 inp["pretrain_aud_feat"])
    return fn


def test_ddim25_plain_beat_matches_reference():
    cfg = get_config("beat")
    f = golden("ddim25_plain_beat.npz")
    inp = make_inputs(cfg, int(f["batch"]), seed=int(f["input_seed"]))
    src = S.NoiseSource(seed=int(f["noise_seed"]))
    tr = []
    x = S.ddim_sample_loop(_eps_fn("beat", inp), (2, cfg.n_poses, cfg.net_dim_pose), {}, src, trace=tr)
    assert src.i == int(f["draws"]) == 26
    for i, (_, k, xs, x0) in enumerate(tr):
        assert k == 24 - i
        np.testing.assert_allclose(xs[:, :3, :6], f["step_corner"][i], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(x0[:, :3, :6], f["x0_corner"][i], rtol=1e-6, atol=1e-6)
    scale = float(np.abs(f["final"]).max())
    assert float((x - torch.from_numpy(f["final"])).abs().max()) <= 1e-6 * scale


def test_harmonize_show_matches_reference():
    cfg = get_config("show", jump_length=3, jump_n_sample=2)
    f = golden("ddim25_harmonize_show_3_2.npz")
    B, L = int(f["batch"]), cfg.overlap_len
    inp = make_inputs(cfg, B, seed=int(f["input_seed"]))
    g = torch.Generator().manual_seed(int(f["gt_seed"]))
    gt = torch.zeros(B, cfg.n_poses, cfg.net_dim_pose)
    gt[:, :L] = torch.randn(B, L, cfg.net_dim_pose, generator=g)
    mask = torch.zeros_like(gt, dtype=torch.bool)
    mask[:, :L] = True
    src = S.NoiseSource(seed=int(f["noise_seed"]))
    x = S.ddim_sample_loop(_eps_fn("show", inp), (B, cfg.n_poses, cfg.net_dim_pose), {"gt": gt, "outpainting_mask": mask},
                           src, jump_length=3, jump_n_sample=2, overlap_len=L)
    assert src.i == int(f["draws"]) == 1 + 27 * 2 + 12
    scale = float(np.abs(f["final"]).max())
    assert float((x - torch.from_numpy(f["final"])).abs().max()) <= 1e-6 * scale


@pytest.mark.parametrize("name", ["ddim25_eta05_show", "ddim25_eta05_masked_show"])
def test_ddim25_eta_matches_reference(name):
    """DDIM with eta != 0 (gaussian_diffusion.py:1011-1032): sigma * randn_like enters every step but the last, the eps
    coefficient becomes sqrt(1 - abar_prev - sigma^2); plain loop and the masked out-painting schedule (the RePaint noise weight
    stays sqrt(1 - abar_prev))."""
    cfg = get_config("show")
    f = golden(f"{name}.npz")
    B, L = int(f["batch"]), cfg.overlap_len
    inp = make_inputs(cfg, B, seed=int(f["input_seed"]))
    y = {}
    if "masked" in name:
        g = torch.Generator().manual_seed(int(f["gt_seed"]))
        gt = torch.zeros(B, cfg.n_poses, cfg.net_dim_pose)
        gt[:, :L] = torch.randn(B, L, cfg.net_dim_pose, generator=g)
        mask = torch.zeros_like(gt, dtype=torch.bool)
        mask[:, :L] = True
        y = {"gt": gt, "outpainting_mask": mask}
    src = S.NoiseSource(seed=int(f["noise_seed"]))
    x = S.ddim_sample_loop(_eps_fn("show", inp), (B, cfg.n_poses, cfg.net_dim_pose), y, src, overlap_len=L, eta=float(f["eta"]))
    assert src.i == int(f["draws"])
    scale = float(np.abs(f["final"]).max())
    assert float((x - torch.from_numpy(f["final"])).abs().max()) <= 2e-6 * scale


def test_fix_head_var_is_a_no_op_of_the_ancestral_loop():
    """opt.fix_head_var in p_sample (gaussian_diffusion.py:759-766) edits a [B, 1, 1] mask with `[..., 90:] = 0`: an empty slice.
    The fixture is the reference's 50-step loop WITH the switch on (the generating script asserts it equals the run without);
    the oracle, which has no such switch, must reproduce it."""
    cfg = get_config("show")
    f = golden("ddpm50_fhv_show.npz")
    assert int(f["identical_to_switch_off"]) == 1
    B = int(f["batch"])
    inp = make_inputs(cfg, B, seed=int(f["input_seed"]))
    src = S.NoiseSource(seed=int(f["noise_seed"]))
    x = S.p_sample_loop(_eps_fn("show", inp), (B, cfg.n_poses, cfg.net_dim_pose), src, n_steps=int(f["diffusion_steps"]))
    assert src.i == int(f["draws"]) == 51
    scale = float(np.abs(f["final"]).max())
    assert float((x - torch.from_numpy(f["final"])).abs().max()) <= 3e-6 * scale


def test_ddpm_prefix_matches_reference():
    """First 40 of the 1000 ancestral steps of BASELINE config 1 (the full loop runs in the GPU suite)."""
    cfg = get_config("beat")
    f = golden("ddpm1000_beat.npz")
    inp = make_inputs(cfg, 1, seed=int(f["input_seed"]))
    src = S.NoiseSource(seed=int(f["noise_seed"]))
    tb = S.diffusion_tables(S.linear_betas(1000))
    fn = _eps_fn("beat", inp)
    x = src.randn((1, cfg.n_poses, cfg.net_dim_pose))
    for i, t in enumerate(range(999, 959, -1)):
        c1, c2 = S._f32(tb["sqrt_recip_alphas_cumprod"], t), S._f32(tb["sqrt_recipm1_alphas_cumprod"], t)
        x, _ = S.ddpm_step(tb, t, x, fn(x, t, c1, c2), src)
        np.testing.assert_allclose(x[:, :3, :6], f["step_corner"][i], rtol=2e-6, atol=2e-6)
        assert abs(float(x.abs().max()) - f["step_stats"][i][2]) <= 2e-6 * f["step_stats"][i][2]


def test_window_chain_tail_matches_reference():
    """3-window chain ending in a 30-frame tail window, through the oracle's window_chain (H2)."""
    cfg = get_config("show")
    f = golden("chain_tail_show.npz")
    N = int(f["frames"])
    inp = make_inputs(cfg, 1, frames=N, seed=int(f["input_seed"]))
    sd = synthetic_sd("show")
    draws = []

    def sample_window(i, a, h, y):
        src = S.NoiseSource(seed=int(f["noise_seed_base"]) + i)

        def fn(x, t, c1, c2):
            with torch.no_grad():
                return D.unidiffuser(sd, cfg, x, torch.full((1,), t), c1, c2, a, inp["person_id"], h)
        out = S.ddim_sample_loop(fn, (1, a.shape[1], cfg.net_dim_pose), y, src, overlap_len=cfg.overlap_len)
        draws.append(src.i)
        return out
    out = S.window_chain(sample_window, inp["audio_emb"], inp["pretrain_aud_feat"], cfg.n_poses, cfg.overlap_len,
                         cfg.net_dim_pose)
    assert draws == list(f["draws"]) and list(f["window_lens"]) == [88, 88, 30]
    scale = float(np.abs(f["out"]).max())
    assert float((out - torch.from_numpy(f["out"])).abs().max()) <= 2e-6 * scale


# ---- round-2 fixtures: BEAT out-painting, sampler switches, the SHOW + CFG ancestral loop -------------------------------
def _masked(cfg, f):
    B, L = int(f["batch"]), cfg.overlap_len
    inp = make_inputs(cfg, B, seed=int(f["input_seed"]))
    g = torch.Generator().manual_seed(int(f["gt_seed"]))
    gt = torch.zeros(B, cfg.n_poses, cfg.net_dim_pose)
    gt[:, :L] = torch.randn(B, L, cfg.net_dim_pose, generator=g)
    mask = torch.zeros_like(gt, dtype=torch.bool)
    mask[:, :L] = True
    return B, inp, {"gt": gt, "outpainting_mask": mask}


def test_harmonize_beat_matches_reference():
    """BEAT out-painting window: overlap_len 4, no CFG, default (3,5) schedule."""
    cfg = get_config("beat")
    f = golden("ddim25_harmonize_beat_3_5.npz")
    B, inp, y = _masked(cfg, f)
    src = S.NoiseSource(seed=int(f["noise_seed"]))
    x = S.ddim_sample_loop(_eps_fn("beat", inp), (B, cfg.n_poses, cfg.net_dim_pose), y, src, overlap_len=cfg.overlap_len)
    assert src.i == int(f["draws"]) == 175
    assert float((x - torch.from_numpy(f["final"])).abs().max()) <= 2e-6 * float(np.abs(f["final"]).max())


def test_no_repaint_and_no_resample_match_reference():
    cfg = get_config("show")
    for name, kw, draws in [("ddim25_norepaint_show.npz", {"no_repaint": True}, 51), ("ddim25_noresample_show.npz", {"no_resample": True}, 31)]:
        f = golden(name)
        B, inp, y = _masked(cfg, f)
        src = S.NoiseSource(seed=int(f["noise_seed"]))
        x = S.ddim_sample_loop(_eps_fn("show", inp), (B, cfg.n_poses, cfg.net_dim_pose), y, src, overlap_len=cfg.overlap_len, **kw)
        assert src.i == int(f["draws"]) == draws, name
        assert float((x - torch.from_numpy(f["final"])).abs().max()) <= 2e-6 * float(np.abs(f["final"]).max()), name


def test_clip_denoised_matches_reference():
    cfg = get_config("show")
    f = golden("ddim25_clip_show.npz")
    B = int(f["batch"])
    inp = make_inputs(cfg, B, seed=int(f["input_seed"]))
    src = S.NoiseSource(seed=int(f["noise_seed"]))
    x = S.ddim_sample_loop(_eps_fn("show", inp), (B, cfg.n_poses, cfg.net_dim_pose), {}, src, clip_denoised=True)
    assert float((x - torch.from_numpy(f["final"])).abs().max()) <= 2e-6


def test_ddpm_show_cfg_prefix_matches_reference():
    """First 12 of the 1000 ancestral steps of the config-5 workload (SHOW, CFG 1.25, B = 2); the full loop runs on the GPU."""
    cfg = get_config("show")
    f = golden("ddpm1000_show.npz")
    B = int(f["batch"])
    inp = make_inputs(cfg, B, seed=int(f["input_seed"]))
    src = S.NoiseSource(seed=int(f["noise_seed"]))
    tb = S.diffusion_tables(S.linear_betas(1000))
    fn = _eps_fn("show", inp)
    x = src.randn((B, cfg.n_poses, cfg.net_dim_pose))
    for i, t in enumerate(range(999, 987, -1)):
        c1, c2 = S._f32(tb["sqrt_recip_alphas_cumprod"], t), S._f32(tb["sqrt_recipm1_alphas_cumprod"], t)
        x, _ = S.ddpm_step(tb, t, x, fn(x, t, c1, c2), src)
        np.testing.assert_allclose(x[:, :3, :6], f["step_corner"][i], rtol=3e-6, atol=3e-6 * f["step_stats"][i][2])


def test_window_chain_beat_tail_matches_reference():
    """BEAT chain (34, 34, 16-frame tail; overlap 4) produced through DDPMTrainer_beat.generate_batch."""
    cfg = get_config("beat")
    f = golden("chain_tail_beat.npz")
    N = int(f["frames"])
    inp = make_inputs(cfg, 1, frames=N, seed=int(f["input_seed"]))
    sd = synthetic_sd("beat")

    def sample_window(i, a, h, y):
        src = S.NoiseSource(seed=int(f["noise_seed_base"]) + i)

        def fn(x, t, c1, c2):
            with torch.no_grad():
                return D.unidiffuser(sd, cfg, x, torch.full((1,), t), c1, c2, a, inp["person_id"], h)
        return S.ddim_sample_loop(fn, (1, a.shape[1], cfg.net_dim_pose), y, src, overlap_len=cfg.overlap_len)
    out = S.window_chain(sample_window, inp["audio_emb"], inp["pretrain_aud_feat"], cfg.n_poses, cfg.overlap_len, cfg.net_dim_pose)
    assert list(f["window_lens"]) == [34, 34, 16]
    assert float((out - torch.from_numpy(f["out"])).abs().max()) <= 2e-6 * float(np.abs(f["out"]).max())

def test_fix_very_first_chain_matches_reference():
    """--fix_very_first (ddpm_show_trainer.py:885-888): window 0 is out-painted too, from the LAST overlap_len frames of the
    first ground-truth window; fixture through DDPMTrainer_show.generate_batch (two masked windows: 175 draws each)."""
    cfg = get_config("show")
    f = golden("chain_fvf_show.npz")
    N = int(f["frames"])
    inp = make_inputs(cfg, 1, frames=N, seed=int(f["input_seed"]))
    motions = torch.randn(1, N, cfg.net_dim_pose, generator=torch.Generator().manual_seed(int(f["motions_seed"])))
    sd = synthetic_sd("show")
    counts = []

    def sample_window(i, a, h, y):
        src = S.NoiseSource(seed=int(f["noise_seed_base"]) + i)

        def fn(x, t, c1, c2):
            with torch.no_grad():
                return D.unidiffuser(sd, cfg, x, torch.full((1,), t), c1, c2, a, inp["person_id"], h)
        out = S.ddim_sample_loop(fn, (1, a.shape[1], cfg.net_dim_pose), y, src, overlap_len=cfg.overlap_len)
        counts.append(src.i)
        return out
    out = S.window_chain(sample_window, inp["audio_emb"], inp["pretrain_aud_feat"], cfg.n_poses, cfg.overlap_len, cfg.net_dim_pose,
                         fix_very_first_motions=motions)
    assert counts == list(f["draws"]) == [175, 175]
    assert torch.equal(out[:, 0], motions[:, cfg.n_poses - cfg.overlap_len])      # frame 0 of the cross-fade is the pinned frame itself
    assert float((out - torch.from_numpy(f["out"])).abs().max()) <= 2e-6 * float(np.abs(f["out"]).max())


def test_same_overlap_noisy_chain_matches_reference():
    """--same_overlap_noisy (gaussian_diffusion.py:1040-1060) through the BEAT harness: windows k > 0 take the previous
    window's saved noisy tail of each level instead of a re-noised gt (no gt-noise draw: 1 + 63 + 48 = 112 draws)."""
    cfg = get_config("beat")
    f = golden("chain_tail_beat_son.npz")
    N = int(f["frames"])
    inp = make_inputs(cfg, 1, frames=N, seed=int(f["input_seed"]))
    sd = synthetic_sd("beat")
    tails, draws = {}, []

    def sample_window(i, a, h, y):
        src = S.NoiseSource(seed=int(f["noise_seed_base"]) + i)

        def fn(x, t, c1, c2):
            with torch.no_grad():
                return D.unidiffuser(sd, cfg, x, torch.full((1,), t), c1, c2, a, inp["person_id"], h)
        out = S.ddim_sample_loop(fn, (1, a.shape[1], cfg.net_dim_pose), y, src, overlap_len=cfg.overlap_len, tails=tails, clip_idx=i)
        draws.append(src.i)
        return out
    out = S.window_chain(sample_window, inp["audio_emb"], inp["pretrain_aud_feat"], cfg.n_poses, cfg.overlap_len, cfg.net_dim_pose)
    assert draws == list(f["draws"]) == [26, 112, 112]
    assert float((out - torch.from_numpy(f["out"])).abs().max()) <= 2e-6 * float(np.abs(f["out"]).max())


# ---- (f)-4: the single-MotionTransformer model (opt.unidiffuser = False, runner.py:46-57) ----------
def test_single_motion_transformer_matches_reference():
    from diffsheg_amd.weights import make_synthetic_state_dict
    cfg = get_config("show", unidiffuser=False)
    f = golden("single_transformer_show.npz")
    sd = make_synthetic_state_dict(cfg, int(f["weight_seed"]))
    assert "joint_embed.weight" in sd and tuple(sd["audio_proj.weight"].shape) == (cfg.aud_latent_dim, cfg.audio_dim)
    B = int(f["batch"])
    inp = make_inputs(cfg, B, seed=int(f["input_seed"]))

    def eps_fn(x, t, c1=None, c2=None, inp=inp):
        with torch.no_grad():
            return D.single_motion_transformer(sd, cfg, x, torch.full((x.shape[0],), int(t), dtype=torch.long), inp["audio_emb"],
                                               inp["person_id"], inp["pretrain_aud_feat"])
    for tag in ("k3", "k20"):
        e = eps_fn(inp["x_T"], int(f[f"{tag}_t"]))
        assert float((e - torch.from_numpy(f[f"{tag}_eps"])).abs().max()) < 2e-5
    src = S.NoiseSource(seed=int(f["noise_seed"]))
    tr = []
    x = S.ddim_sample_loop(eps_fn, (B, cfg.n_poses, cfg.net_dim_pose), {}, src, trace=tr)
    assert src.i == int(f["draws"]) == 26
    for i, (_, k, xs, _x0) in enumerate(tr):
        np.testing.assert_allclose(xs[:, :3, :6], f["step_corner"][i], rtol=2e-6, atol=2e-6)
    scale = float(np.abs(f["final"]).max())
    assert float((x - torch.from_numpy(f["final"])).abs().max()) <= 2e-6 * scale
