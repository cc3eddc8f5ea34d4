"""Yardstick only (never on the product path): what torch.matmul (hipBLASLt / rocBLAS) reaches on the Linear shapes of the
SHOW B=950 CFG step, same box, same clocks — the bare GEMM without LayerNorm / FiLM / activation / residual epilogues."""
import torch
M, Mc = 167200, 83600
shapes = [("q|k|v", M, 1536, 512), ("sa proj_out", M, 512, 512), ("ffn.linear1", M, 1024, 512), ("ffn.linear2", M, 512, 1024),
          ("feat_proj.1", Mc, 1024, 1024), ("feat_proj.3", Mc, 512, 1024), ("square 8192 (peak check)", 8192, 8192, 8192)]
for name, m, n, k in shapes:
    A = torch.randn(m, k, device="cuda").bfloat16(); W = (torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16()
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(5): torch.matmul(A, W.t(), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): torch.matmul(A, W.t(), out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print(f"{name:26s} M={m:6d} N={n:4d} K={k:4d}: {us:8.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TFLOP/s")
