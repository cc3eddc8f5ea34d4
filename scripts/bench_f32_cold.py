"""fp32 GEMM launches of configs[1] back to back on the SAME operands (everything warm in L2 / MALL) vs cycling through 24 operand sets
(> 600 MB: weights, rows and outputs come from HBM, as between the layers of an evaluation)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsheg_amd import _lib
L = _lib.lib()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
M, dev, NSET = 8704, "cuda", 24
for name, n, k in [("ffn.linear2", 512, 1024), ("q|k|v", 1536, 512), ("feat_proj.1", 1024, 960)]:
    sets = [(torch.randn(M, k, device=dev), torch.randn(n, k, device=dev) / k ** 0.5, torch.randn(n, device=dev), torch.empty(M, n, device=dev)) for _ in range(NSET)]
    for label, cyc in (("warm", False), ("cold", True)):
        for pro, kern in ((None, "gemm_nt"), (0, "pro0")):
            def run(i):
                A, W, b, Cc = sets[i % NSET if cyc else 0]
                if pro is None: _lib.check(L.dsh_op_gemm(None, 0, P(A), P(W), P(b), None, P(Cc), None, M, n, k, 0))
                else: _lib.check(L.dsh_op_gemm_f32_pro(None, 0, P(A), k, k, None, 0, 0, None, 0, 0, None, 0, 0, k, P(W), P(b), None, None, 0, 0, 1, 1, None, P(Cc), M, n, 0, None, 0, None))
            for i in range(NSET): run(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(4 * NSET): run(i)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000 / (4 * NSET)
            print(f"{name:12s} N={n:5d} K={k:5d} {kern:8s} {label}: {us:7.1f} us  {2.0 * M * n * k / us / 1e6:6.1f} TF/s")
