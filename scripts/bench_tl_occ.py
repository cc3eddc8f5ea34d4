import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsheg_amd import _lib
L = _lib.lib(); P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
dev = "cuda"; n = 1536
W = (torch.randn(n, 512, device=dev) / 22).bfloat16(); b = torch.randn(n, device=dev)
for blocks in [128, 256, 512, 768, 1024, 2048]:
    M = blocks * 128
    X = torch.randn(M, 512, device=dev).bfloat16(); Ct = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    def run(): _lib.check(L.dsh_op_tl_linear(None, 0, P(X), P(W), P(b), None, None, P(Ct), M, n, 0, None, None, None, 88, 1, 512))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"blocks={blocks:5d}: {us:8.1f} us  {2.0*M*n*512/us/1e6:7.1f} TF/s")
