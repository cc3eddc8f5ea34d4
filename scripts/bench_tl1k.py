import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsheg_amd import _lib
L = _lib.lib(); P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
dev = "cuda"
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
K = 1024
for name, Mv, n, act, res, cf, ct in [("ffn2", 167200, 512, 0, False, False, True), ("f1", 83600, 1024, 1, False, False, True), ("f3", 83600, 512, 0, True, True, True)]:
    M = (Mv + 127) // 128 * 128
    torch.manual_seed(0)
    X = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(n, K, device=dev) / K ** 0.5).bfloat16()
    b = torch.randn(n, device=dev); R = torch.randn(M, n, device=dev) if res else None
    Cf = torch.empty(M, n, device=dev) if cf else None; Ct = torch.empty(M, n, device=dev, dtype=torch.bfloat16) if ct else None
    def run(): _lib.check(L.dsh_op_tl_linear(None, 0, P(X), P(W), P(b), P(R), P(Cf), P(Ct), Mv, n, act, None, None, None, 88, 1, K))
    us = timeit(run)
    print(f"TL1K {name:5s} N={n}: {us:8.1f} us  {2.0*Mv*n*K/us/1e6:7.1f} TF/s")
    rows = torch.cat([torch.arange(0, 200), torch.arange(Mv - 150, Mv)]).to(dev)
    ref = X[rows].float() @ W.float().T + b
    ref = {0: lambda v: v, 1: torch.nn.functional.silu, 2: torch.nn.functional.gelu}[act](ref)
    if res: ref = ref + R[rows]
    got = Cf[rows] if cf else Ct[rows].float()
    print("      max err:", (got - ref).abs().max().item(), " ref max", ref.abs().max().item())
