"""Micro-benchmark of the hot GEMM shapes of the SHOW B=950 CFG config (M = 167200 / 83600 tokens)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsheg_amd import _lib
L = _lib.lib()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
M, Mc = 167200, 83600
shapes = [("qkv", M, 1536, 512, 0, False, False, True), ("sty", M, 512, 512, 0, True, True, True),
          ("ffn1", M, 1024, 512, 2, False, False, True), ("ffn2", M, 512, 1024, 0, False, False, True),
          ("f1", Mc, 1024, 896, 1, False, False, True), ("f3", Mc, 512, 1024, 0, True, True, False)]
dev = "cuda"
for name, m, n, k, act, res, cf, ct in shapes:
    A = torch.randn(m, k, device=dev).bfloat16(); W = (torch.randn(n, k, device=dev) / k ** 0.5).bfloat16()
    b = torch.randn(n, device=dev); R = torch.randn(m, n, device=dev) if res else None
    Cf = torch.empty(m, n, device=dev) if cf else None; Ct = torch.empty(m, n, device=dev, dtype=torch.bfloat16) if ct else None
    def run():
        _lib.check(L.dsh_op_gemm(None, 1, P(A), P(W), P(b), P(R), P(Cf), P(Ct), m, n, k, act))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = 2.0 * m * n * k
    byt = m * k * 2 + n * k * 2 + (m * n * 4 if res else 0) + (m * n * 4 if cf else 0) + (m * n * 2 if ct else 0)
    print(f"{name:5s} M={m} N={n} K={k}: {us:8.1f} us  {fl/us/1e6:7.1f} TF/s  {byt/us/1e3:6.2f} GB/s(min traffic {byt/1e6:.0f} MB)")
    # correctness spot check on a few rows
    ref = (A[:64].float() @ W.float().T + b)
    ref = {0: lambda v: v, 1: torch.nn.functional.silu, 2: torch.nn.functional.gelu}[act](ref)
    if res: ref = ref + R[:64]
    got = Cf[:64] if cf else Ct[:64].float()
    print("      max err first rows:", (got - ref).abs().max().item())
