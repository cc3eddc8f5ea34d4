"""Round 6: the LDS-tiled kernel class (tl4.hip: weights AND activations through an LDS ring, 64 x 128 outputs per wave) against the
register-stationary kernels (tl2.hip) it replaces at whole-chip token counts.  The arithmetic is the same operation for operation, so
every output bit must agree — which also keeps the window-chain kernels (tl_small.hip) and the sharded bit-identity invariants intact.
Reference ops: models/transformer.py:284-289,304-338 (feat_proj), :119-125 (q | k | v behind one LayerNorm)."""
import ctypes as C

import pytest
import torch

from diffsheg_amd import _lib

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr())


def _run(monkeypatch, tl4, variant, pro, X, W, b, R, Mv, N, act, gam, bet, kreal, K, hilo):
    monkeypatch.setenv("DSH_TL2", "1")
    monkeypatch.setenv("DSH_HILO", "1" if hilo else "0")
    monkeypatch.setenv("DSH_TL4", "7" if tl4 else "0")
    monkeypatch.setenv("DSH_TL4_MIN_ROWS", "0")
    monkeypatch.setenv("DSH_TL4_V", variant)
    d = X.device
    Cf = torch.full((Mv, N), float("nan"), device=d) if R is not None else None
    Ct = torch.full((Mv, N), float("nan"), device=d, dtype=torch.bfloat16)
    P = lambda t: None if t is None else _p(t)
    L = _lib.lib()
    _lib.check(L.dsh_op_tl_linear(None, pro, _p(X), _p(W), _p(b), P(R), P(Cf), _p(Ct), Mv, N, act, _p(gam), _p(bet), None,
                                  kreal if pro == 3 else 88, 1, K))
    torch.cuda.synchronize()
    fam = L.dsh_debug_last_tl_variant()
    return fam, Ct.view(torch.int16).cpu(), (None if Cf is None else Cf.view(torch.int32).cpu())


# (K, N, pro, act, residual planes, real concat width); token rows: several rounds of 128- and 256-token tiles with a ragged last one
CASES = [(1024, 1024, 3, 1, False, 999), (1024, 1024, 3, 1, False, 896), (1024, 512, 0, 0, True, 1024), (512, 1536, 1, 0, False, 512)]


@pytest.mark.parametrize("K,N,pro,act,res,kreal", CASES)
@pytest.mark.parametrize("variant", ["a", "b"])
def test_lds_tiled_kernels_are_bit_identical_to_the_register_stationary_ones(K, N, pro, act, res, kreal, variant, monkeypatch):
    Mv = (256 * 130 + 77) if K == 512 else (256 * 70 + 77)     # >= 128 token blocks of the tl2 kernels (no N split), ragged last tile
    d = "cuda:0"
    g = torch.Generator().manual_seed(K + N + pro + kreal)
    X = torch.randn(Mv, K, generator=g) * 1.5 + 0.3
    W = torch.randn(N, K, generator=g) / K ** 0.5
    gam = 1 + 0.1 * torch.randn(K, generator=g); bet = 0.1 * torch.randn(K, generator=g)
    if pro == 3:
        X[:, kreal:] = 0; W[:, kreal:] = 0; gam[kreal:] = 0; bet[kreal:] = 0
    X, W, gam, bet = X.bfloat16().to(d), W.bfloat16().to(d), gam.to(d), bet.to(d)
    b = torch.randn(N, generator=g).to(d)
    R = torch.randn(Mv, N, generator=g).to(d) if res else None
    fam0, ct0, cf0 = _run(monkeypatch, False, variant, pro, X, W, b, R, Mv, N, act, gam, bet, kreal, K, res)
    fam1, ct1, cf1 = _run(monkeypatch, True, variant, pro, X, W, b, R, Mv, N, act, gam, bet, kreal, K, res)
    assert fam0 in (1, 2), fam0                          # the rolling tl2 kernels (whole-chip path: no N split at >= 128 token blocks)
    assert fam1 == (4 if variant == "a" else 5), fam1    # ... and the LDS-tiled kernel really ran
    assert torch.isfinite(ct1.view(torch.bfloat16).float()).all()
    assert torch.equal(ct0, ct1)
    if res:
        assert torch.equal(cf0, cf1)
    # and against fp64 on the same rounded operands (all rows)
    xin = X[:, :kreal].double()
    if pro in (1, 3):
        xin = torch.nn.functional.layer_norm(xin, (kreal,), gam[:kreal].double(), bet[:kreal].double(), 1e-5)
    ref = xin @ W[:, :kreal].double().T + b.double()
    if act == 1:
        ref = torch.nn.functional.silu(ref)
    if res:
        ref = ref + R.double()
        got = cf1.view(torch.float32).double()
    else:
        got = ct1.view(torch.bfloat16).double()
    err = ((got - ref.cpu()).abs().max() / ref.abs().max()).item()
    print(f"[tl4 {variant} K={K} N={N} pro={pro} kreal={kreal}] max err / range vs fp64: {err:.2e}")
    assert err < (2e-3 if res else 1.2e-2)


@pytest.mark.parametrize("K,N,pro,act,res,kreal", CASES + [(512, 512, 2, 0, True, 512)])
def test_rotated_weight_stream_order_changes_no_bit(K, N, pro, act, res, kreal, monkeypatch):
    """Round 6: DSH_TL2_ROT — block b of a rolling tl2 launch starts its weight stream at tile (b / 8) % tiles instead of tile 0 (the
    32 CUs of an XCD no longer ask their L2 for the same weight lines at the same moment).  Tiles are independent: every bit must agree."""
    Mv, T, nb = (256 * 130 + 77) if K == 512 else (128 * 130 + 50), 88, 400
    d = "cuda:0"
    g = torch.Generator().manual_seed(K + N + pro + kreal + 1)
    X = torch.randn(Mv, K, generator=g) * 1.5 + 0.3
    W = torch.randn(N, K, generator=g) / K ** 0.5
    gam = 1 + 0.1 * torch.randn(K, generator=g); bet = 0.1 * torch.randn(K, generator=g)
    if pro == 3:
        X[:, kreal:] = 0; W[:, kreal:] = 0; gam[kreal:] = 0; bet[kreal:] = 0
    X, W, gam, bet = X.bfloat16().to(d), W.bfloat16().to(d), gam.to(d), bet.to(d)
    b = torch.randn(N, generator=g).to(d)
    R = torch.randn(Mv, N, generator=g).to(d) if res else None
    film = (0.3 * torch.randn(nb, 2 * K, generator=g)).to(d) if pro == 2 else None
    monkeypatch.setenv("DSH_TL2", "1"); monkeypatch.setenv("DSH_TL4", "0"); monkeypatch.setenv("DSH_HILO", "1" if res else "0")
    outs = {}
    P = lambda t: None if t is None else _p(t)
    for rot in ("0", "1"):
        monkeypatch.setenv("DSH_TL2_ROT", rot)
        Cf = torch.full((Mv, N), float("nan"), device=d) if res else None
        Ct = torch.full((Mv, N), float("nan"), device=d, dtype=torch.bfloat16)
        _lib.check(_lib.lib().dsh_op_tl_linear(None, pro, _p(X), _p(W), _p(b), P(R), P(Cf), _p(Ct), Mv, N, act, _p(gam), _p(bet), P(film),
                                               kreal if pro == 3 else T, nb, K))
        torch.cuda.synchronize()
        assert _lib.lib().dsh_debug_last_tl_variant() in (1, 2)
        outs[rot] = (Ct.view(torch.int16).cpu(), None if Cf is None else Cf.view(torch.int32).cpu())
    assert torch.isfinite(outs["1"][0].view(torch.bfloat16).float()).all()
    assert torch.equal(outs["0"][0], outs["1"][0])
    if res:
        assert torch.equal(outs["0"][1], outs["1"][1])


@pytest.mark.parametrize("kreal", [896, 999])
def test_skipping_the_zero_padded_k_steps_changes_no_bit(kreal, monkeypatch):
    """Round 6: the rolling feat_proj.1 kernel does not multiply the trailing all-zero fragments of the concat row (8 for the expression
    encoder's 896 columns, 1 for the gesture encoder's 999; DSH_TL2_KSKIP=0: all 64 fragments as before)."""
    K = N = 1024
    Mv = 128 * 130 + 50
    d = "cuda:0"
    g = torch.Generator().manual_seed(kreal)
    X = torch.randn(Mv, K, generator=g) * 1.5 + 0.3
    W = torch.randn(N, K, generator=g) / K ** 0.5
    gam = 1 + 0.1 * torch.randn(K, generator=g); bet = 0.1 * torch.randn(K, generator=g)
    X[:, kreal:] = 0; W[:, kreal:] = 0; gam[kreal:] = 0; bet[kreal:] = 0
    X, W, gam, bet = X.bfloat16().to(d), W.bfloat16().to(d), gam.to(d), bet.to(d)
    b = torch.randn(N, generator=g).to(d)
    monkeypatch.setenv("DSH_TL2", "1"); monkeypatch.setenv("DSH_TL4", "0"); monkeypatch.setenv("DSH_HILO", "0")
    outs = {}
    for sw in ("0", "1"):
        monkeypatch.setenv("DSH_TL2_KSKIP", sw)
        Ct = torch.full((Mv, N), float("nan"), device=d, dtype=torch.bfloat16)
        _lib.check(_lib.lib().dsh_op_tl_linear(None, 3, _p(X), _p(W), _p(b), None, None, _p(Ct), Mv, N, 1, _p(gam), _p(bet), None, kreal, 1, K))
        torch.cuda.synchronize()
        assert _lib.lib().dsh_debug_last_tl_variant() == 1
        outs[sw] = Ct.view(torch.int16).cpu()
    assert torch.isfinite(outs["1"].view(torch.bfloat16).float()).all()
    assert torch.equal(outs["0"], outs["1"])
