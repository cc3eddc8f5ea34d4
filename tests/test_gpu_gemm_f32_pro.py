"""fp32 parity path, round 6: the Linears that carry their LayerNorm / StylizationBlock front inside the launch
(diffsheg_amd/csrc/gemm_f32_pro.hip) vs an fp64 restatement of models/transformer.py:86-97 (StylizationBlock),
:119-125 (sa_block.norm + q|k|v) and :284-289, :304-312 (feat_proj.0 LayerNorm over the concat + feat_proj.1),
called through the C ABI (dsh_op_gemm_f32_pro); and the whole evaluation with the fused launches against the
separate row kernels of rounds 1 - 5 (DSH_F32_FUSE=0)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from diffsheg_amd import _lib  # noqa: E402
from diffsheg_amd.config import get_config  # noqa: E402
from diffsheg_amd.synthetic import make_inputs  # noqa: E402
from util import max_abs, synthetic_sd  # noqa: E402


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _fold(W, b, gamma, beta, Kp):
    """LN(x) W^T + b = rstd (x W'^T - mean c) + d with W' = gamma (.) W (zero padded to Kp columns), c = W' 1, d = b + W beta."""
    N, K = W.shape
    Wf = torch.zeros(N, Kp, dtype=torch.float64)
    Wf[:, :K] = W.double() * gamma.double()
    return Wf.float(), Wf.float().double().sum(1).float(), (b.double() + W.double() @ beta.double()).float()


@pytest.mark.parametrize("M,N,widths,k_real,act,offset", [
    (8704, 1536, (512, 0, 0, 0), 512, 0, 0.0),          # q|k|v of configs[1] (BEAT, 256 clips x 34 frames)
    (1000 + 37, 1024, (512, 256, 128, 64), 947, 1, 0.0),     # feat_proj.1 of the BEAT gesture encoder: concat, 13 zero-padded columns, ragged M
    (300, 1024, (512, 256, 128, 128), 999, 1, 0.0),     # SHOW gesture encoder
    (130, 1024, (512, 256, 128, 0), 896, 1, 0.0),       # expression encoder: three segments
    (200, 512, (512, 0, 0, 0), 512, 0, 50.0),           # large row mean: the shifted one-pass moments
    (200, 1024, (512, 256, 128, 64), 947, 1, -200.0),
])
def test_folded_layernorm_linear_matches_fp64(M, N, widths, k_real, act, offset):
    g = torch.Generator().manual_seed(M + N + k_real)
    K = sum(widths)
    segs, col = [], 0
    for w in widths:
        if w == 0:
            segs.append(None)
            continue
        real = max(0, min(w, k_real - col))
        x = torch.zeros(M, w + 32)                      # leading dimension wider than the segment: rows are strided
        x[:, :real] = torch.randn(M, real, generator=g) * 1.7 + offset
        segs.append(x)
        col += w
    X = torch.cat([s[:, :w] for s, w in zip(segs, widths) if s is not None], 1)[:, :k_real]
    W = torch.randn(N, k_real, generator=g) / k_real ** 0.5
    b, gamma, beta = torch.randn(N, generator=g), 1 + 0.3 * torch.randn(k_real, generator=g), 0.3 * torch.randn(k_real, generator=g)
    ref = torch.nn.functional.layer_norm(X.double(), (k_real,), gamma.double(), beta.double(), 1e-5) @ W.double().T + b.double()
    if act == 1:
        ref = torch.nn.functional.silu(ref)
    Wf, fc, fd = _fold(W, b, gamma, beta, K)
    d = "cuda:0"
    sd = [s.to(d) if s is not None else None for s in segs]
    Wd, fcd, fdd = Wf.to(d), fc.to(d), fd.to(d)
    out = torch.full((M, N), float("nan"), device=d)
    args = []
    for s, w in zip(sd, widths):
        args += [_p(s), (w + 32) if s is not None else 0, w]
    _lib.check(_lib.lib().dsh_op_gemm_f32_pro(None, 1, *args, k_real, _p(Wd), _p(fdd), _p(fcd), None, 0, 0, 1, 1, None, _p(out), M, N, act, None, 0, None))
    torch.cuda.synchronize()
    err = (out.cpu().double() - ref).abs().max().item()
    # |mean| / std = 30 .. 120 at the offsets: x W' and mean c cancel to ~1e-5 relative of |offset| sqrt(K) — the fold's own fp32 limit, inside the
    # path's 1e-3 gate; plain rows sit at fp32 summation round-off
    tol = 3e-5 * max(1.0, ref.abs().max().item()) * (1.0 if offset == 0.0 else 20.0)
    print(f"[fold M={M} N={N} K={K} P={k_real} offset={offset}] max err {err:.2e} (|ref| max {ref.abs().max().item():.2f})")
    assert err < tol, err


@pytest.mark.parametrize("M,N,K,frames,nb,offset", [(8704, 512, 512, 34, 256, 0.0), (2 * 264 + 5, 512, 512, 88, 3, 0.0), (77, 128, 128, 11, 7, 0.0),
                                                    (300, 512, 512, 30, 10, 40.0)])
def test_stylization_block_linear_matches_fp64(M, N, K, frames, nb, offset):
    g = torch.Generator().manual_seed(M + K)
    y = torch.randn(M, K, generator=g) * 2.0 + offset
    gamma, beta = 1 + 0.3 * torch.randn(K, generator=g), 0.3 * torch.randn(K, generator=g)
    scale, shift = 0.5 * torch.randn(nb, K, generator=g), 0.5 * torch.randn(nb, K, generator=g)
    W, b = torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g)
    clip = (torch.arange(M) // frames) % nb
    hn = torch.nn.functional.layer_norm(y.double(), (K,), gamma.double(), beta.double(), 1e-5)
    ref = torch.nn.functional.silu(hn * (1 + scale.double()[clip]) + shift.double()[clip]) @ W.double().T + b.double() + R.double()
    # the folded per-clip table the denoiser builds with film_expand_kernel (fold = 1), at a column offset inside a wider row
    off = 4 * K
    film = torch.zeros(nb, off + 2 * K + 8)
    film[:, off:off + K] = (gamma.double() * (1 + scale.double())).float()
    film[:, off + K:off + 2 * K] = (beta.double() * (1 + scale.double()) + shift.double()).float()
    d = "cuda:0"
    yd, Wd, bd, Rd, fd = y.to(d), W.to(d), b.to(d), R.to(d), film.to(d)
    out = Rd.clone()                                    # in place, as the residual stream is updated
    _lib.check(_lib.lib().dsh_op_gemm_f32_pro(None, 2, _p(yd), K, K, None, 0, 0, None, 0, 0, None, 0, 0, K, _p(Wd), _p(bd), None, _p(fd), film.shape[1], off,
                                              frames, nb, _p(out), _p(out), M, N, 0, None, 0, None))
    torch.cuda.synchronize()
    err = (out.cpu().double() - ref).abs().max().item()
    print(f"[sty M={M} N={N} K={K}] max err {err:.2e} (|ref| max {ref.abs().max().item():.2f})")
    assert err < 3e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("M,offset", [(8704, 0.0), (300 + 11, 30.0)])
def test_row_moments_travel_from_the_producer_to_the_stylization_launch(M, offset):
    """ffn.linear2 (pro 0) leaves the group moments of the rows it writes; the StylizationBlock launch behind it (pro 2) combines them instead of
    reading its input twice (models/transformer.py:172-181, :86-97).  Checked against fp64 and against the launch that takes its own moments."""
    g = torch.Generator().manual_seed(M)
    K1, D, frames, nb = 1024, 512, 34, 9
    a = torch.randn(M, K1, generator=g)
    W2, b2 = torch.randn(D, K1, generator=g) / K1 ** 0.5, torch.randn(D, generator=g) + offset
    gamma, beta = 1 + 0.3 * torch.randn(D, generator=g), 0.3 * torch.randn(D, generator=g)
    scale, shift = 0.5 * torch.randn(nb, D, generator=g), 0.5 * torch.randn(nb, D, generator=g)
    W3, b3 = torch.randn(D, D, generator=g) / D ** 0.5, torch.randn(D, generator=g)
    R = torch.randn(M, D, generator=g)
    clip = (torch.arange(M) // frames) % nb
    y2 = a.double() @ W2.double().T + b2.double()
    hn = torch.nn.functional.layer_norm(y2, (D,), gamma.double(), beta.double(), 1e-5)
    ref = torch.nn.functional.silu(hn * (1 + scale.double()[clip]) + shift.double()[clip]) @ W3.double().T + b3.double() + R.double()
    film = torch.cat([(gamma.double() * (1 + scale.double())).float(), (beta.double() * (1 + scale.double()) + shift.double()).float()], 1).contiguous()
    d = "cuda:0"
    ad, W2d, b2d, W3d, b3d, fd = a.to(d), W2.to(d), b2.to(d), W3.to(d), b3.to(d), film.to(d)
    y2d = torch.empty(M, D, device=d)
    st = torch.full((M, D // 32, 2), float("nan"), device=d)
    L = _lib.lib()
    _lib.check(L.dsh_op_gemm_f32_pro(None, 0, _p(ad), K1, K1, None, 0, 0, None, 0, 0, None, 0, 0, K1, _p(W2d), _p(b2d), None, None, 0, 0, 1, 1, None, _p(y2d), M, D, 0,
                                     None, 0, _p(st)))
    outs = []
    for use in (True, False):
        out = R.to(d)
        _lib.check(L.dsh_op_gemm_f32_pro(None, 2, _p(y2d), D, D, None, 0, 0, None, 0, 0, None, 0, 0, D, _p(W3d), _p(b3d), None, _p(fd), 2 * D, 0, frames, nb, _p(out), _p(out),
                                         M, D, 0, _p(st) if use else None, D // 32 if use else 0, None))
        torch.cuda.synchronize()
        outs.append(out.cpu().double())
    # the group moments themselves
    yg = y2d.cpu().double().view(M, D // 32, 32)
    assert (st[..., 0].cpu().double() - yg.mean(-1)).abs().max().item() < 1e-5 * max(1.0, abs(offset))
    assert ((st[..., 1].cpu().double() - ((yg - yg.mean(-1, keepdim=True)) ** 2).sum(-1)).abs() / 32).max().item() < 1e-4
    e_ref, e_ab = (outs[0] - ref).abs().max().item(), (outs[0] - outs[1]).abs().max().item()
    print(f"[moments side channel M={M} offset={offset}] vs fp64 {e_ref:.2e}, vs own-moments launch {e_ab:.2e}")
    assert e_ref < 5e-5 * max(1.0, ref.abs().max().item()) and e_ab < 2e-5 * max(1.0, ref.abs().max().item())


def test_lds_dma_and_register_staging_give_identical_results(tmp_path):
    """The operands reach LDS by DMA (default) or through the staging registers (DSH_GP_DMA=0): the arithmetic is the same operation for
    operation — including the order in which the row moments are accumulated — so the outputs are bit-identical (fresh processes: the switches
    are read once).  The StylizationBlock launch with specialised producer waves (DSH_GP_WS=1; measured slower, off by default) accumulates its
    row moments in another order: fp32 round-off apart."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for dma, ws in (("1", "0"), ("0", "0"), ("1", "1")):
        f = str(tmp_path / f"gp_dma_{dma}_{ws}.pt")
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "gp_dma_worker.py"), f], env=dict(os.environ, DSH_GP_DMA=dma, DSH_GP_WS=ws), capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0 and "GP_DMA_WORKER_OK" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
        res.append(torch.load(f))
    for k in ("pro0", "pro1", "pro2"):
        assert torch.isfinite(res[0][k]).all()
        assert torch.equal(res[0][k], res[1][k]), (k, float((res[0][k] - res[1][k]).abs().max()))
    assert torch.equal(res[0]["pro0"], res[2]["pro0"]) and torch.equal(res[0]["pro1"], res[2]["pro1"])
    d = float((res[0]["pro2"] - res[2]["pro2"]).abs().max())
    print(f"[StylizationBlock launch, producer waves vs four-wave form] max |d| = {d:.2e}")
    assert d < 2e-5 * max(1.0, float(res[0]["pro2"].abs().max())), d


@pytest.mark.parametrize("ds,B,T", [("beat", 16, 34), ("beat", 256, 34), ("show", 8, 88), ("show", 21, 30)])
def test_fused_fronts_agree_with_the_row_kernels(ds, B, T, monkeypatch):
    """The same evaluation with the LayerNorm / StylizationBlock fronts inside the GEMM launches (default above 512 token rows, the few-row GEMM's
    range) and as the separate row kernels (DSH_F32_FUSE=0): fp32 round-off apart (folded affine, one-pass moments, hardware exp / rcp in SiLU),
    far inside the 1e-3 gate.  SHOW runs with classifier-free guidance: the CFG-null constant of layer l + 1 is handed over in layer l's last epilogue."""
    from diffsheg_amd.model import UniDiffuser
    cfg = get_config(ds)
    inp = make_inputs(cfg, B, frames=T, seed=77)
    outs = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("DSH_F32_FUSE", fuse)
        model = UniDiffuser(cfg, synthetic_sd(ds), device="cuda:0", precision="fp32")
        shape_e = (B, T, cfg.expression_dim)
        eps = model(inp["x_T"].cuda(), torch.full((B,), 420, dtype=torch.long).cuda(), sqrt_alphas=[torch.full(shape_e, 1.3), torch.full(shape_e, 0.8)],
                    audio_emb=inp["audio_emb"].cuda(), length=torch.full((B,), T), person_id=inp["person_id"].cuda(),
                    add_cond={"pretrain_aud_feat": inp["pretrain_aud_feat"].cuda()}, pe_type="pe_sinu", y={})
        # (debug taps exist on single-stream evaluations only: the 256-clip batch is split over sub-batch streams)
        outs.append((eps.cpu(), model.debug_tap("expr_x0").cpu() if B <= 16 else eps[..., cfg.split_pos:].cpu()))
        del model
    e, ex = max_abs(outs[0][0], outs[1][0]), max_abs(outs[0][1], outs[1][1])
    print(f"[f32 fuse on / off {ds} B={B}] max |d eps| = {e:.2e}, max |d expr_x0| = {ex:.2e}")
    assert e < 1e-4 and ex < 1e-4, (e, ex)
