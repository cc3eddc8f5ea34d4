"""Do an HBM-bound TL launch (StylizationBlock) and an MFMA-bound one (ffn.linear2 / q|k|v) overlap when issued on two
streams over half the rows each?  Compares serial full-row launches with concurrent half-row launches."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsheg_amd import _lib
L = _lib.lib()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
dev = "cuda"
os.environ["DSH_TL_RAW"] = "1"
Mfull = 167200; Mh = 83712; Mp = 167424 + 512; T = 88; nb = 950
def mk(n, K, res, cf, seed):
    torch.manual_seed(seed)
    X = (torch.randn(Mp, K, device=dev) * 1.5 + 0.3).bfloat16(); W = (torch.randn(n, K, device=dev) / K ** 0.5).bfloat16()
    b = torch.randn(n, device=dev); R = torch.randn(Mp, n, device=dev) if res else None
    gam = 1 + 0.1 * torch.randn(K, device=dev); bet = 0.1 * torch.randn(K, device=dev)
    film = 0.3 * torch.randn(nb * 2, 1024, device=dev)
    Cf = torch.empty(Mp, n, device=dev) if cf else None; Ct = torch.empty(Mp, n, device=dev, dtype=torch.bfloat16)
    return dict(X=X, W=W, b=b, R=R, gam=gam, bet=bet, film=film, Cf=Cf, Ct=Ct, n=n, K=K)
def launch(a, pro, act, M, stream):
    _lib.check(L.dsh_op_tl_linear(C.c_void_p(stream.cuda_stream), pro, P(a["X"]), P(a["W"]), P(a["b"]), P(a["R"]), P(a["Cf"]), P(a["Ct"]),
                                  M, a["n"], act, P(a["gam"]), P(a["bet"]), P(a["film"]), T, nb * 2, a["K"]))
sty = mk(512, 512, True, True, 0); ffn2 = mk(512, 1024, False, False, 1); qkv = mk(1536, 512, False, False, 2)
sty2 = mk(512, 512, True, True, 3); ffn2b = mk(512, 1024, False, False, 4); qkvb = mk(1536, 512, False, False, 5)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def wall(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for nameB, B1, B2, proB in [("ffn2", ffn2, ffn2b, 0), ("qkv", qkv, qkvb, 1)]:
    def serial():
        launch(sty, 2, 0, Mfull, s1); launch(B1, proB, 0, Mfull, s1)
    def concurrent():   # the same total work as two half-row pairs: stream 1 runs sty then B, stream 2 runs B then sty
        launch(sty, 2, 0, Mh, s1); launch(B1, proB, 0, Mh, s1)
        launch(B2, proB, 0, Mh, s2); launch(sty2, 2, 0, Mh, s2)
    print(f"sty + {nameB}: serial full rows {wall(serial):8.1f} us   two streams, half rows each, opposite order {wall(concurrent):8.1f} us", flush=True)
