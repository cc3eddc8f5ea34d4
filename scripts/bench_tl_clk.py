"""Shader clock under load: run a TL kernel back-to-back, then read the in-kernel clock probe of the last launch."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsheg_amd import _lib
L = _lib.lib()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
dev = "cuda"
os.environ["DSH_TL_RAW"] = "1"
Mv = 167200; M = (Mv + 255) // 256 * 256 + 256; T = 88; nb = 950
for name, n, act, res, cf, pro in [("qkv", 1536, 0, False, False, 1), ("sty", 512, 0, True, True, 2), ("ffn1", 1024, 2, False, False, 0)]:
    torch.manual_seed(0)
    X = (torch.randn(M, 512, device=dev) * 1.5 + 0.3).bfloat16(); W = (torch.randn(n, 512, device=dev) / 512 ** 0.5).bfloat16()
    b = torch.randn(n, device=dev); R = torch.randn(M, n, device=dev) if res else None
    gam = 1 + 0.1 * torch.randn(512, device=dev); bet = 0.1 * torch.randn(512, device=dev)
    film = 0.3 * torch.randn(nb * 2, 1024, device=dev)
    Cf = torch.empty(M, n, device=dev) if cf else None; Ct = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    def run():
        _lib.check(L.dsh_op_tl_linear(None, pro, P(X), P(W), P(b), P(R), P(Cf), P(Ct), Mv, n, act, P(gam), P(bet), P(film), T, nb * 2, 512))
    for reps in (50,):
        os.environ["DSH_TL_DBG"] = str(64 << 8) if pro in (1, 2) else "0"
        os.environ["DSH_TL_CLK"] = "1"
        for _ in range(reps): run()
        os.environ["DSH_TL_CLK"] = "2"
        print(name, "after", reps, "back-to-back launches:", flush=True)
        run()
        torch.cuda.synchronize()
