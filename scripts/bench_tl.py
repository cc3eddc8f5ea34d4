"""Micro-benchmark + correctness of the token-per-lane fused Linear (K=512) vs the tiled GEMM."""
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
pros = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
Mv = 167200; M = (Mv + 127) // 128 * 128; T = 88; nb = 950   # buffers padded to the 128-row block
for name, n, act, res, cf, ct, pro in [("qkv", 1536, 0, False, False, True, 1), ("sty", 512, 0, True, True, True, 2), ("ffn1", 1024, 2, False, False, True, 0)]:
    if pro not in pros and 0 not in pros: continue
    use_pro = pro if pro in pros else 0
    torch.manual_seed(0)
    X = (torch.randn(M, 512, device=dev) * 1.5 + 0.3).bfloat16(); W = (torch.randn(n, 512, device=dev) / 512 ** 0.5).bfloat16()
    b = torch.randn(n, device=dev); R = torch.randn(M, n, device=dev) if res else None
    gam = 1 + 0.1 * torch.randn(512, device=dev); bet = 0.1 * torch.randn(512, device=dev)
    film = 0.3 * torch.randn(nb * 2, 1024, device=dev)
    Cf = torch.empty(M, n, device=dev) if cf else None; Ct = torch.empty(M, n, device=dev, dtype=torch.bfloat16) if ct else None
    def run():
        _lib.check(L.dsh_op_tl_linear(None, use_pro, P(X), P(W), P(b), P(R), P(Cf), P(Ct), Mv, n, act, P(gam), P(bet), P(film), T, nb * 2, 512))
    us = timeit(run)
    fl = 2.0 * Mv * n * 512
    print(f"TL {name:5s} pro={use_pro} N={n}: {us:8.1f} us  {fl/us/1e6:7.1f} TF/s")
    rows = torch.cat([torch.arange(0, 200), torch.arange(Mv - 150, Mv)]).to(dev)
    xin = X[rows].float()
    if use_pro >= 1:
        xin = torch.nn.functional.layer_norm(xin, (512,), gam, bet, 1e-5)
    if use_pro == 2:
        f = film[(rows // T) % (nb * 2)]
        xin = torch.nn.functional.silu(xin * (1 + f[:, :512]) + f[:, 512:])
    ref = xin.bfloat16().float() @ W.float().T + b
    ref = {0: lambda v: v, 1: torch.nn.functional.silu, 2: torch.nn.functional.gelu}[act](ref)
    if res: ref = ref + R[rows]
    got = Cf[rows] if cf else Ct[rows].float()
    print("      max err sample rows:", (got - ref).abs().max().item(), " ref max", ref.abs().max().item())
