"""Timing ablations of the two dominant token-per-lane instantiations (results of ablated runs are garbage by design)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsheg_amd import _lib
L = _lib.lib()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
dev = "cuda"
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
Mv = 167200; M = (Mv + 127) // 128 * 128 + 128; T = 88; nb = 950
NAMES = {1: "no-barrier", 2: "no-store", 4: "no-ldsread", 8: "no-Wload", 16: "no-mfma"}
abls = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 0, 1, 2, 4, 8, 16, 3, 9, 13, 15, 27, 31, 29]
os.environ["DSH_TL_RAW"] = "1"
for name, n, res, cf, pro in [("qkv", 1536, False, False, 1), ("sty", 512, True, True, 2)]:
    torch.manual_seed(0)
    X = (torch.randn(M, 512, device=dev) * 1.5 + 0.3).bfloat16(); W = (torch.randn(n, 512, device=dev) / 512 ** 0.5).bfloat16()
    b = torch.randn(n, device=dev); R = torch.randn(M, n, device=dev) if res else None
    gam = 1 + 0.1 * torch.randn(512, device=dev); bet = 0.1 * torch.randn(512, device=dev)
    film = 0.3 * torch.randn(nb * 2, 1024, device=dev)
    Cf = torch.empty(M, n, device=dev) if cf else None; Ct = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    def run():
        _lib.check(L.dsh_op_tl_linear(None, pro, P(X), P(W), P(b), P(R), P(Cf), P(Ct), Mv, n, 0, P(gam), P(bet), P(film), T, nb * 2, 512))
    for abl in abls:
        os.environ["DSH_TL_DBG"] = str(abl << 8)
        us = timeit(run)
        fl = 2.0 * Mv * n * 512
        desc = "+".join(v for k, v in NAMES.items() if abl & k) or "full"
        print(f"TL {name:4s} abl={abl:2d} {desc:45s}: {us:8.1f} us  {fl/us/1e6:7.1f} TF/s-equiv", flush=True)
os.environ["DSH_TL_DBG"] = "0"
