"""Worker of tests/test_gpu_gemm_f32_pro.py::test_lds_dma_and_register_staging_give_identical_results: runs the three launch kinds of
gemm_f32_pro.hip on seeded operands and writes the outputs to argv[1] (DSH_GP_DMA is read once per process)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from diffsheg_amd import _lib  # noqa: E402

P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
L = _lib.lib()
g = torch.Generator().manual_seed(11)
M, d = 1000 + 37, "cuda:0"
outs = {}
# pro 0: feat_proj.3-like, residual in place
A, W, b, R = torch.randn(M, 1024, generator=g), torch.randn(512, 1024, generator=g) / 32, torch.randn(512, generator=g), torch.randn(M, 512, generator=g)
Ad, Wd, bd, o = A.to(d), W.to(d), b.to(d), R.to(d)
_lib.check(L.dsh_op_gemm_f32_pro(None, 0, P(Ad), 1024, 1024, None, 0, 0, None, 0, 0, None, 0, 0, 1024, P(Wd), P(bd), None, None, 0, 0, 1, 1, P(o), P(o), M, 512, 0, None, 0, None))
outs["pro0"] = o.cpu()
# pro 1: four concat segments, 13 padded columns
segs = [torch.randn(M, w, generator=g) for w in (512, 256, 128, 64)]
segs[3][:, 51:] = 0
W1, b1, fc = torch.randn(1024, 960, generator=g) / 31, torch.randn(1024, generator=g), torch.randn(1024, generator=g)
W1[:, 947:] = 0
sd = [x.to(d) for x in segs]
W1d, b1d, fcd = W1.to(d), b1.to(d), fc.to(d)
o1 = torch.empty(M, 1024, device=d)
_lib.check(L.dsh_op_gemm_f32_pro(None, 1, P(sd[0]), 512, 512, P(sd[1]), 256, 256, P(sd[2]), 128, 128, P(sd[3]), 64, 64, 947, P(W1d), P(b1d), P(fcd), None, 0, 0, 1, 1, None, P(o1),
                                 M, 1024, 1, None, 0, None))
outs["pro1"] = o1.cpu()
# pro 2
y, film = torch.randn(M, 512, generator=g) * 2, torch.randn(31, 1024, generator=g)
W2, b2 = torch.randn(512, 512, generator=g) / 23, torch.randn(512, generator=g)
yd, fd, W2d, b2d, o2 = y.to(d), film.to(d), W2.to(d), b2.to(d), R.to(d)
_lib.check(L.dsh_op_gemm_f32_pro(None, 2, P(yd), 512, 512, None, 0, 0, None, 0, 0, None, 0, 0, 512, P(W2d), P(b2d), None, P(fd), 1024, 0, 34, 31, P(o2), P(o2), M, 512, 0, None, 0, None))
outs["pro2"] = o2.cpu()
torch.cuda.synchronize()
torch.save(outs, sys.argv[1])
print("GP_DMA_WORKER_OK", os.environ.get("DSH_GP_DMA"))
