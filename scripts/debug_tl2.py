import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsheg_amd import _lib
L = _lib.lib(); P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
d = "cuda"
cases = [(512, 1536, 1, 0, False, False, True), (512, 1024, 0, 2, False, False, True), (1024, 512, 0, 0, False, False, True), (1024, 1024, 0, 1, False, False, True),
         (512, 512, 2, 0, True, True, True), (1024, 512, 0, 0, True, True, True), (512, 256, 0, 0, False, True, False)]
for Mv in (40000, 167200):
  for (K, N, pro, act, res, cf, ct) in cases:
    M = (Mv + 255) // 256 * 256
    g = torch.Generator().manual_seed(1)
    X = (torch.randn(M, K, generator=g)).bfloat16().to(d); W = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().to(d)
    b = torch.randn(N, generator=g).to(d); R = torch.randn(M, N, generator=g).to(d) if res else None
    gam = (1 + 0.1 * torch.randn(K, generator=g)).to(d); bet = (0.1 * torch.randn(K, generator=g)).to(d)
    nb = 7; T = 88
    film = (0.3 * torch.randn(nb, 2 * K, generator=g)).to(d)
    Cf = torch.full((M, N), float("nan"), device=d) if cf else None
    Ct = torch.full((M, N), float("nan"), device=d, dtype=torch.bfloat16) if ct else None
    for rep in range(3):
        _lib.check(L.dsh_op_tl_linear(None, pro, P(X), P(W), P(b), P(R), P(Cf), P(Ct), Mv, N, act, P(gam), P(bet), P(film), T, nb, K))
        torch.cuda.synchronize()
        rows = torch.arange(Mv, device=d)
        xin = X[:Mv].float()
        if pro >= 1: xin = torch.nn.functional.layer_norm(xin, (K,), gam, bet, 1e-5)
        if pro == 2:
            f = film[(rows // T) % nb]; xin = torch.nn.functional.silu(xin * (1 + f[:, :K]) + f[:, K:])
        ref = xin.bfloat16().float() @ W.float().T + b
        ref = {0: lambda v: v, 1: torch.nn.functional.silu, 2: torch.nn.functional.gelu}[act](ref)
        if res: ref = ref + R[:Mv]
        got = (Cf if cf else Ct)[:Mv].float()
        err = (got - ref).abs()
        bad = err > 0.1
        nbad = int(bad.sum().item())
        msg = f"K={K} N={N} pro={pro} act={act} res={res} M={Mv} rep {rep}: max err {err.max().item():.3e} bad {nbad}"
        if nbad:
            idx = bad.nonzero()
            msg += f" | bad rows blk256 {sorted(set((idx[:, 0] // 256).tolist()))[:8]} col tiles {sorted(set((idx[:, 1] // 32).tolist()))[:12]} first {idx[0].tolist()}"
        print(msg, flush=True)
    del X, W, R, Cf, Ct
