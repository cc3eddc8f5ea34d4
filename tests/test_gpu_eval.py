"""Boundary-1 parity on a real MI355X: dsh_eval (through diffsheg_amd.model.UniDiffuser) vs
  (a) golden fixtures produced by the imported reference (tests/golden/eval_*.npz), and
  (b) the CPU oracle on fresh seeded inputs / odd shapes (tail-window T, B=1, per-sample t).
Tolerance (north_star): 1e-3 absolute on eps (|eps| ~ 3.5) for the fp32 path; the bf16 path is
reported against the same vectors with a loose gate."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from diffsheg_amd.config import get_config  # noqa: E402
from diffsheg_amd.synthetic import make_inputs  # noqa: E402
from oracle import denoiser_ref  # noqa: E402
from util import golden, gpu_model, max_abs, synthetic_sd  # noqa: E402

FP32_ATOL = 1e-3
# bf16 path, one evaluation, |eps| ~ 3.5: ~3x the error measured on MI355X (max 1.9e-2, rms 5e-3; printed by the tests)
BF16_MAX, BF16_RMS = 6e-2, 1.5e-2


def _call(model, cfg, inp, t, c1, c2):
    B, T = inp["x_T"].shape[:2]
    shape_e = (B, T, cfg.expression_dim)
    c1t = c1 if torch.is_tensor(c1) else torch.full((B,), float(c1))
    c2t = c2 if torch.is_tensor(c2) else torch.full((B,), float(c2))
    sa = [c1t.view(B, 1, 1).expand(shape_e), c2t.view(B, 1, 1).expand(shape_e)]
    tt = t if torch.is_tensor(t) else torch.full((B,), int(t), dtype=torch.long)
    return model(inp["x_T"].cuda(), tt.cuda(), sqrt_alphas=sa, audio_emb=inp["audio_emb"].cuda(),
                 length=torch.full((B,), T), person_id=inp["person_id"].cuda(),
                 add_cond={"pretrain_aud_feat": inp["pretrain_aud_feat"].cuda()}, pe_type="pe_sinu", y={})


@pytest.mark.parametrize("ds", ["beat", "show"])
def test_eval_fp32_matches_reference_golden(ds):
    cfg = get_config(ds)
    f = golden(f"eval_{ds}.npz")
    model = gpu_model(ds, "fp32")
    inp = make_inputs(cfg, int(f["batch"]), seed=int(f["input_seed"]))
    worst = 0.0
    for tag in ["k0", "k14", "k24", "t999", "t1"]:
        eps = _call(model, cfg, inp, int(f[f"{tag}_t"]), float(f[f"{tag}_c1"]), float(f[f"{tag}_c2"]))
        ref = torch.from_numpy(f[f"{tag}_eps"])
        e = max_abs(eps, ref)
        worst = max(worst, e)
        assert e < FP32_ATOL, (tag, e)
        eps_exp = eps[..., cfg.split_pos:]
        assert max_abs(eps_exp, torch.from_numpy(f[f"{tag}_eps_exp"])) < FP32_ATOL
        if tag == "k14":
            assert max_abs(model.debug_tap("aud_feat"), torch.from_numpy(f["k14_aud_feat"])) < FP32_ATOL
    print(f"[eval fp32 {ds}] worst |eps - ref| = {worst:.3e}")


@pytest.mark.parametrize("ds,B,T", [("show", 3, 30), ("show", 1, 88), ("beat", 5, 34), ("show", 2, 11)])
def test_eval_fp32_matches_oracle_odd_shapes(ds, B, T):
    cfg = get_config(ds)
    sd = synthetic_sd(ds)
    model = gpu_model(ds, "fp32")
    inp = make_inputs(cfg, B, frames=T, seed=21 + B)
    t = torch.tensor([(37 * i + 5) % 1000 for i in range(B)])      # per-sample timesteps
    c1 = 1.0 + torch.arange(B, dtype=torch.float32)
    c2 = 0.5 + 0.25 * torch.arange(B, dtype=torch.float32)
    eps = _call(model, cfg, inp, t, c1, c2)
    with torch.no_grad():
        ref, parts = denoiser_ref.unidiffuser(sd, cfg, inp["x_T"], t, c1.view(B, 1, 1), c2.view(B, 1, 1), inp["audio_emb"],
                                              inp["person_id"], inp["pretrain_aud_feat"], return_parts=True)
    assert max_abs(model.debug_tap("expr_x0"), parts["expr_x0"]) < 2e-3 * max(1.0, float(c1.max()))
    assert max_abs(eps, ref) < FP32_ATOL


def test_eval_cfg_scale_one_disables_doubling():
    """cond_scale == 1 with classifier_free weights: no CFG doubling (transformer.py:537)."""
    cfg = get_config("show", cond_scale=1.0)
    model = gpu_model("show", "fp32", cond_scale=1.0)
    inp = make_inputs(cfg, 2, frames=20, seed=9)
    eps = _call(model, cfg, inp, 400, 1.5, 1.1)
    with torch.no_grad():
        ref = denoiser_ref.unidiffuser(synthetic_sd("show"), cfg, inp["x_T"], torch.full((2,), 400), torch.tensor(1.5),
                                       torch.tensor(1.1), inp["audio_emb"], inp["person_id"], inp["pretrain_aud_feat"])
    assert max_abs(eps, ref) < FP32_ATOL


def test_eval_is_deterministic_and_recomputes_on_new_condition():
    cfg = get_config("show")
    model = gpu_model("show", "fp32")
    a = make_inputs(cfg, 2, frames=24, seed=1)
    b = make_inputs(cfg, 2, frames=24, seed=2)
    e1 = _call(model, cfg, a, 100, 2.0, 1.7)
    e2 = _call(model, cfg, b, 100, 2.0, 1.7)
    e3 = _call(model, cfg, a, 100, 2.0, 1.7)
    assert torch.equal(e1, e3)
    assert not torch.equal(e1, e2)


@pytest.mark.parametrize("ds", ["beat", "show"])
def test_eval_bf16_close_to_reference(ds):
    cfg = get_config(ds)
    f = golden(f"eval_{ds}.npz")
    model = gpu_model(ds, "bf16")
    inp = make_inputs(cfg, int(f["batch"]), seed=int(f["input_seed"]))
    eps = _call(model, cfg, inp, int(f["k14_t"]), float(f["k14_c1"]), float(f["k14_c2"]))
    ref = torch.from_numpy(f["k14_eps"])
    e = max_abs(eps, ref)
    rms = float((eps.cpu() - ref).pow(2).mean().sqrt())
    print(f"[eval bf16 {ds}] max|eps-ref| = {e:.3e}, rms = {rms:.3e} (|eps| max {float(ref.abs().max()):.2f})")
    assert e < BF16_MAX and rms < BF16_RMS  # reported and gated at ~3x the measured error (the 1e-3 bar is the fp32 path's)


@pytest.mark.parametrize("B,T", [(1, 15), (2, 11), (3, 30), (5, 88), (2, 120)])
def test_eval_bf16_short_and_long_windows_match_oracle(B, T):
    """Tail windows of the chain can be as short as overlap_len + 1 frames and n_poses is a free option (T = 120 takes the
    row-major attention fallback): the token-per-lane path (FiLM rows of up to six
    clips staged per 128-token block) must handle them at chain batch sizes."""
    cfg = get_config("show")
    sd = synthetic_sd("show")
    model = gpu_model("show", "bf16")
    inp = make_inputs(cfg, B, frames=T, seed=31 + B)
    t = torch.tensor([(37 * i + 5) % 1000 for i in range(B)])
    c1 = 1.0 + torch.arange(B, dtype=torch.float32)
    c2 = 0.5 + 0.25 * torch.arange(B, dtype=torch.float32)
    eps = _call(model, cfg, inp, t, c1, c2)
    with torch.no_grad():
        ref = denoiser_ref.unidiffuser(sd, cfg, inp["x_T"], t, c1.view(B, 1, 1), c2.view(B, 1, 1), inp["audio_emb"], inp["person_id"],
                                       inp["pretrain_aud_feat"])
    e = max_abs(eps, ref)
    rms = float((eps.cpu() - ref).pow(2).mean().sqrt())
    print(f"[eval bf16 show B={B} T={T}] max|eps-ref| = {e:.3e}, rms = {rms:.3e}")
    assert e < BF16_MAX * max(1.0, float(c1.max()) / 2) and rms < BF16_RMS * max(1.0, float(c1.max()) / 2)


def _big_batch(cfg, B, seed):
    """B distinct clips at full window length with per-clip timesteps / coefficients."""
    inp = make_inputs(cfg, B, seed=seed)
    t = torch.tensor([(37 * i + 5) % 1000 for i in range(B)])
    c1 = 1.0 + 0.5 * (torch.arange(B, dtype=torch.float32) % 7) / 7
    c2 = 0.5 + 0.25 * (torch.arange(B, dtype=torch.float32) % 5) / 5
    return inp, t, c1, c2


def _oracle_rows(ds, cfg, inp, t, c1, c2, rows):
    r = torch.tensor(rows)
    with torch.no_grad():
        return denoiser_ref.unidiffuser(synthetic_sd(ds), cfg, inp["x_T"][r], t[r], c1[r].view(-1, 1, 1), c2[r].view(-1, 1, 1),
                                        inp["audio_emb"][r], inp["person_id"][r], inp["pretrain_aud_feat"][r])


def test_eval_bf16_headline_batch_sampled_clips_match_oracle():
    """BASELINE config 3 at its REAL size (SHOW, B = 950, T = 88, CFG: 167 200 token rows, three sub-batch streams): clips
    are independent, so the oracle is evaluated on a sample of them only — first / last clip of the batch (= of each CFG
    half), both sides of the stream split points (clips 316 | 317 and 633 | 634; 475 for a two-stream run), clips that
    straddle 128-token block boundaries (88-frame clips: every one after the first), and a few interior ones."""
    cfg = get_config("show")
    B = 950
    model = gpu_model("show", "bf16")
    inp, t, c1, c2 = _big_batch(cfg, B, seed=41)
    eps = _call(model, cfg, inp, t, c1, c2).cpu()
    assert torch.isfinite(eps).all()
    rows = [0, 1, 2, 3, 315, 316, 317, 474, 475, 632, 633, 634, 700, 948, 949]      # 1: tokens 88..175 straddle block 128; stream splits
    ref = _oracle_rows("show", cfg, inp, t, c1, c2, rows)
    got = eps[torch.tensor(rows)]
    worst = 0.0
    for j, r in enumerate(rows):
        e = float((got[j] - ref[j]).abs().max()); rms = float((got[j] - ref[j]).pow(2).mean().sqrt())
        worst = max(worst, e)
        assert e < BF16_MAX and rms < BF16_RMS, (r, e, rms)
    print(f"[eval bf16 show B=950] sampled clips {rows}: worst max|eps-ref| = {worst:.3e}")
    # the un-sampled clips: same magnitude statistics as the sampled ones (catches garbage rows without an oracle eval)
    per_clip = eps.abs().mean(dim=(1, 2))
    assert float(per_clip.max()) < 3.0 * float(per_clip.median()) and float(per_clip.min()) > 0.3 * float(per_clip.median())


def test_eval_fp32_config2_batch_sampled_clips_match_oracle():
    """BASELINE config 2 at its real size (BEAT, B = 256, T = 34, fp32, the <= 1e-3 parity configuration)."""
    cfg = get_config("beat")
    B = 256
    model = gpu_model("beat", "fp32")
    inp, t, c1, c2 = _big_batch(cfg, B, seed=43)
    eps = _call(model, cfg, inp, t, c1, c2).cpu()
    rows = [0, 1, 3, 4, 127, 128, 200, 254, 255]             # 34-frame clips: 3|4 straddles token 128
    ref = _oracle_rows("beat", cfg, inp, t, c1, c2, rows)
    e = max_abs(eps[torch.tensor(rows)], ref)
    print(f"[eval fp32 beat B=256] sampled clips: max|eps-ref| = {e:.3e}")
    assert e < FP32_ATOL
    per_clip = eps.abs().mean(dim=(1, 2))
    assert float(per_clip.max()) < 3.0 * float(per_clip.median()) and float(per_clip.min()) > 0.3 * float(per_clip.median())


@pytest.mark.parametrize("precision,B", [("bf16", 380), ("fp32", 380), ("bf16", 930)])
def test_large_batch_two_stream_split_is_bit_identical(precision, B, monkeypatch):
    """Batches of >= 12288 token rows are evaluated as two (>= 64500: three) sub-batches on as many streams (denoiser.hip,
    DualDenoiser); clips are independent, so the result must equal the single-stream evaluation bit for bit.
    B = 380: 33 440 rows, split 190 | 190 with CFG doubling inside; B = 930: 81 840 rows, three streams of 310."""
    from diffsheg_amd.model import UniDiffuser
    cfg = get_config("show")
    T = 88
    inp = make_inputs(cfg, 8, frames=T, seed=5)
    rep = lambda v: v.repeat(B // 8 + 1, *([1] * (v.dim() - 1)))[:B].contiguous()
    inp = {k: rep(v) for k, v in inp.items()}
    inp["x_T"] = inp["x_T"] + 0.01 * torch.arange(B, dtype=torch.float32).view(B, 1, 1)
    t = torch.tensor([(37 * i + 5) % 1000 for i in range(B)])
    c1 = 1.0 + 0.01 * torch.arange(B, dtype=torch.float32)
    c2 = 0.5 + 0.005 * torch.arange(B, dtype=torch.float32)
    outs = []
    for dual in ("0", "3"):                                   # "3": at most three streams (the default)
        monkeypatch.setenv("DSH_DUAL", dual)
        model = UniDiffuser(cfg, synthetic_sd("show"), device="cuda:0", precision=precision)
        outs.append(_call(model, cfg, inp, t, c1, c2).clone())
        del model
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("ds,B,T", [("show", 1, 88), ("show", 3, 88), ("show", 2, 11), ("show", 5, 40), ("beat", 4, 34), ("show", 12, 88),
                                    ("show", 30, 88), ("beat", 70, 34), ("show", 40, 88)])
def test_small_batch_kernels_are_bit_identical(ds, B, T, monkeypatch):
    """Window-chain batches (<= 6144 token rows per launch) run the token-per-lane Linears as 32-token row blocks with one tile per
    wave (tl_small.hip) instead of N-split 128 / 256-token blocks.  The arithmetic is the same operation for operation, so an
    evaluation must not change by a bit with DSH_TLS=0 — i.e. a clip's result does not depend on the kernel family its batch size
    selected (the sharded long-audio path relies on that: a chain sampled alone equals its row of a batched run).  Covers CFG
    (two row ranges) and no CFG (BEAT), short tail windows with several clips per 32-token block, B = 12 (33 row blocks per half),
    B = 30 (5456 rows), BEAT B = 70 (34-frame clips straddling every row block) and B = 40 (7296 rows: whole-chip kernels for the
    CFG-doubled launches, small ones for the 3520-row conditional half)."""
    from diffsheg_amd.model import UniDiffuser
    cfg = get_config(ds)
    inp = make_inputs(cfg, B, frames=T, seed=77 + B)
    t = torch.tensor([(53 * i + 11) % 1000 for i in range(B)])
    c1 = 1.0 + 0.1 * torch.arange(B, dtype=torch.float32)
    c2 = 0.5 + 0.05 * torch.arange(B, dtype=torch.float32)
    outs = []
    for tls in ("0", "1"):
        monkeypatch.setenv("DSH_TLS", tls)
        model = UniDiffuser(cfg, synthetic_sd(ds), device="cuda:0", precision="bf16")
        outs.append(_call(model, cfg, inp, t, c1, c2).clone())
        del model
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])


def test_fused_ffn_launch_matches_separate_launches(monkeypatch):
    """tl2_ffn_kernel (ffn.linear1 -> GELU -> ffn.linear2 -> StylizationBlock -> + h in one launch) vs the same branch as three
    launches (DSH_FFN_FUSE=0): same operands; the fused kernel keeps the hidden layer and y2 in fp32 registers where the
    separate launches round them to bf16 in HBM, so the two agree to bf16 round-off, not bit for bit."""
    from diffsheg_amd.model import UniDiffuser
    cfg = get_config("show")
    B, T = 192, 88                                            # M = r0 + Mc = 33 920 token rows: the fused launch is active
    inp = make_inputs(cfg, 8, frames=T, seed=6)
    rep = lambda v: v.repeat(B // 8 + 1, *([1] * (v.dim() - 1)))[:B].contiguous()
    inp = {k: rep(v) for k, v in inp.items()}
    inp["x_T"] = inp["x_T"] + 0.01 * torch.arange(B, dtype=torch.float32).view(B, 1, 1)
    t = torch.tensor([(91 * i + 3) % 1000 for i in range(B)])
    c1 = 1.0 + 0.01 * torch.arange(B, dtype=torch.float32)
    c2 = 0.5 + 0.005 * torch.arange(B, dtype=torch.float32)
    monkeypatch.setenv("DSH_DUAL", "0")
    outs = []
    for fuse in ("0", "1"):
        monkeypatch.setenv("DSH_FFN_FUSE", fuse)
        model = UniDiffuser(cfg, synthetic_sd("show"), device="cuda:0", precision="bf16")
        outs.append(_call(model, cfg, inp, t, c1, c2).clone())
        del model
    assert torch.isfinite(outs[0]).all()
    d = (outs[0] - outs[1]).abs()
    print(f"[fused FFN vs separate launches] max {float(d.max()):.3e} rms {float(d.pow(2).mean().sqrt()):.3e}")
    assert float(d.max()) < BF16_MAX and float(d.pow(2).mean().sqrt()) < BF16_RMS


def test_bad_arguments_raise():
    from diffsheg_amd import _lib
    cfg = get_config("show")
    model = gpu_model("show", "fp32")
    inp = make_inputs(cfg, 2, frames=16, seed=4)
    with pytest.raises(ValueError):
        _call(model, cfg, {**inp, "pretrain_aud_feat": inp["pretrain_aud_feat"][:, :8]}, 10, 1.0, 1.0)
    with pytest.raises(ValueError):
        model(inp["x_T"].cuda(), torch.zeros(2, dtype=torch.long), sqrt_alphas=None, audio_emb=inp["audio_emb"],
              length=None, person_id=inp["person_id"], add_cond={"pretrain_aud_feat": inp["pretrain_aud_feat"]})
    with pytest.raises(NotImplementedError):
        model(inp["x_T"].cuda(), torch.zeros(2, dtype=torch.long), sqrt_alphas=[torch.ones(2), torch.ones(2)],
              audio_emb=inp["audio_emb"], length=None, person_id=inp["person_id"],
              add_cond={"pretrain_aud_feat": inp["pretrain_aud_feat"]}, pe_type="learnable")


def test_fused_attention_branch_stage_is_bit_identical_to_the_separate_launch(monkeypatch):
    """Round 5: at whole-chip token counts the StylizationBlock of the attention branch runs as the first stage of the fused FFN launch
    (tl3_ffn_kernel<..., STY>; DSH_FFN_STY, read when a context is created).  Same arithmetic operation for operation as the separate
    launch it replaces (tl2_linear_kernel<512, 2, ..., ROLL, HL>): a whole UniDiffuser evaluation at B = 100 (17 600 token rows: fused FFN,
    two sub-batch streams) must agree BIT FOR BIT between the two forms."""
    import torch
    from diffsheg_amd.config import get_config
    from diffsheg_amd.model import UniDiffuser
    from diffsheg_amd.synthetic import make_inputs
    from util import synthetic_sd
    cfg = get_config("show")
    B = 100
    inp = make_inputs(cfg, B, seed=21)
    t = torch.full((B,), 560, dtype=torch.long)
    shape_e = (B, cfg.n_poses, cfg.expression_dim)
    outs = {}
    for sty in ("0", "1"):
        monkeypatch.setenv("DSH_FFN_STY", sty)
        model = UniDiffuser(cfg, synthetic_sd("show"), device="cuda:0", precision="bf16")
        eps = model(inp["x_T"].cuda(), t.cuda(), sqrt_alphas=[torch.full(shape_e, 4.9), torch.full(shape_e, 4.8)], audio_emb=inp["audio_emb"].cuda(),
                    length=None, person_id=inp["person_id"].cuda(), add_cond={"pretrain_aud_feat": inp["pretrain_aud_feat"].cuda()}, pe_type="pe_sinu", y={})
        torch.cuda.synchronize()
        outs[sty] = eps.cpu()
        del model
    assert torch.isfinite(outs["1"]).all()
    assert torch.equal(outs["0"], outs["1"]), float((outs["0"] - outs["1"]).abs().max())


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_embeddings_on_distinct_rows_change_nothing(prec, monkeypatch):
    """Round 6: with one timestep for the whole batch the time / speaker / FiLM embedding Linears (transformer.py:446-457, :77) run on the
    DISTINCT speakers' rows only and are expanded per clip (DSH_EMB_DEDUP=0: every clip's row as before).  Up to 16 rows both paths
    take the same skinny-GEMM kernel, row by row: bit-identical.  Mixed speakers, repeated speakers, and a per-sample-t batch (which
    must take the per-clip path whatever the switch says) are covered; the oracle pins the values."""
    cfg = get_config("show")
    sd = synthetic_sd("show")
    model = gpu_model("show", prec)
    B, T = 7, 24
    inp = make_inputs(cfg, B, frames=T, seed=77)
    pid = inp["person_id"].clone()
    pid[3] = pid[0]; pid[5] = pid[1]; pid[6] = pid[0]            # 4 distinct speakers among 7 clips
    inp["person_id"] = pid
    outs = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("DSH_EMB_DEDUP", sw)
        outs[sw] = _call(model, cfg, inp, 520, 1.7, 1.3).cpu()
    assert torch.equal(outs["0"], outs["1"])
    with torch.no_grad():
        ref = denoiser_ref.unidiffuser(sd, cfg, inp["x_T"], torch.full((B,), 520), torch.tensor(1.7), torch.tensor(1.3), inp["audio_emb"],
                                       inp["person_id"], inp["pretrain_aud_feat"])
    e = max_abs(outs["1"], ref)
    print(f"[distinct-row embeddings {prec}] max |eps - oracle| = {e:.3e}")
    assert e < (FP32_ATOL if prec == "fp32" else BF16_MAX)
    # per-sample timesteps: the hint is off, every clip is its own row
    monkeypatch.setenv("DSH_EMB_DEDUP", "1")
    t = torch.tensor([(91 * i + 3) % 1000 for i in range(B)])
    eps = _call(model, cfg, inp, t, 1.7, 1.3).cpu()
    with torch.no_grad():
        ref = denoiser_ref.unidiffuser(sd, cfg, inp["x_T"], t, torch.tensor(1.7), torch.tensor(1.3), inp["audio_emb"], inp["person_id"],
                                       inp["pretrain_aud_feat"])
    assert max_abs(eps, ref) < (FP32_ATOL if prec == "fp32" else BF16_MAX)


def test_fused_encoder_aud_tail_matches_reference_tap(monkeypatch):
    """Round 6: encoder_aud behind its attention as ONE token-per-lane launch (tl_aud.hip; bf16 path): the aud_feat tap against the
    reference's own encoder_aud output (eval_show.npz, k14) and against the six-launch path it replaces (DSH_AUD_FUSE=0)."""
    cfg = get_config("show")
    f = golden("eval_show.npz")
    model = gpu_model("show", "bf16")
    inp = make_inputs(cfg, int(f["batch"]), seed=int(f["input_seed"]))
    want = torch.from_numpy(f["k14_aud_feat"])
    taps = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("DSH_AUD_FUSE", sw)
        _call(model, cfg, inp, int(f["k14_t"]), float(f["k14_c1"]), float(f["k14_c2"]))
        taps[sw] = model.debug_tap("aud_feat").cpu()
    scale = float(want.abs().max())
    e1, e0, d = max_abs(taps["1"], want) / scale, max_abs(taps["0"], want) / scale, max_abs(taps["1"], taps["0"]) / scale
    print(f"[encoder_aud tail] fused vs reference {e1:.2e}, six launches vs reference {e0:.2e}, fused vs six launches {d:.2e} (of range {scale:.2f})")
    assert e1 < 2e-2 and e1 < 1.5 * e0 + 2e-3


@pytest.mark.parametrize("B,T", [(3, 88), (2, 30)])
def test_fused_output_head_is_bit_identical(B, T, monkeypatch):
    """Round 6: `out` of both CFG halves + CFG mix + expression x0 (+ its tiled bf16 copy) in one launch (tl_out.hip) performs the arithmetic of
    the three launches it replaces operation for operation (bias-seeded accumulators in ascending k, the same rounded mix / x0 expressions):
    the whole evaluation — the gesture half sees the expression x0 through the tiled copy — must agree bit for bit (DSH_OUT_FUSE=1 turns the fused launch on; it is off by default: measured slower)."""
    cfg = get_config("show")
    model = gpu_model("show", "bf16")
    inp = make_inputs(cfg, B, frames=T, seed=5 + B)
    outs = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("DSH_OUT_FUSE", sw)
        outs[sw] = _call(model, cfg, inp, 360, 2.1, 1.9).cpu()
    assert torch.isfinite(outs["1"]).all()
    assert torch.equal(outs["0"], outs["1"])


@pytest.mark.parametrize("switch", ["DSH_JOINT_FUSE", "DSH_APROJ_TL", "DSH_AUD_HOIST"])
def test_round6_token_per_lane_launches_agree_with_the_launches_they_replace(switch, monkeypatch):
    """Round 6: the layer-0 seed (joint_embed + PE + CFG-null constant + plane split, tl_embed.hip) and audio_proj as token-per-lane
    launches, and encoder_aud's front computed once per condition, against the GEMM + row-kernel sequences of round 5 (switch = 0) on the
    bf16 path: the same bf16 operands, fp32 accumulation in ascending k, the same epilogue expressions — the whole evaluation agrees to
    fp32 round-off of the intermediate tensors (the seed and the hoist: bit for bit)."""
    cfg = get_config("show")
    model = gpu_model("show", "bf16")
    inp = make_inputs(cfg, 3, frames=88, seed=41)
    outs = {}
    for sw in ("1", "0"):
        monkeypatch.setenv(switch, sw)
        outs[sw] = _call(model, cfg, inp, 720, 3.3, 3.1).cpu()
    d = max_abs(outs["0"], outs["1"])
    print(f"[{switch}] max |eps(new) - eps(round-5 launches)| = {d:.3e}")
    assert torch.isfinite(outs["1"]).all()
    if switch == "DSH_APROJ_TL":
        # tl_aproj seeds its accumulators with the bias, the GEMM adds it behind the k loop: fp32 round-off that flips the bf16 rounding of a
        # few audio_proj outputs; one such flip moves eps by up to ~2e-2 on this path (measured 1.97e-2) — inside the bf16 evaluation gate
        assert d < BF16_MAX
    else:
        assert torch.equal(outs["0"], outs["1"])
