import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsheg_amd import _lib
L = _lib.lib()
def P(t): return C.c_void_p(t.data_ptr())
def run(A, W, dtype=0):
    M, K = A.shape; N = W.shape[0]
    out = torch.full((M, N), float('nan'), device='cuda')
    _lib.check(L.dsh_op_gemm(None, dtype, P(A), P(W), None, None, P(out), None, M, N, K, 0))
    torch.cuda.synchronize()
    return out
torch.manual_seed(0)
for (M, N, K) in [(32, 32, 32), (128, 128, 32), (128, 128, 64), (300, 200, 96)]:
    A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda')
    out = run(A, W); ref = A.double() @ W.double().T
    err = (out.double() - ref).abs()
    print(M, N, K, 'max err', err.max().item(), 'frac bad', (err > 1e-3).float().mean().item())
    if err.max() > 1e-3:
        bad = (err > 1e-3)
        print(' bad rows', bad.any(1).nonzero().flatten()[:40].tolist())
        print(' bad cols', bad.any(0).nonzero().flatten()[:40].tolist())
        # is it a transposition or k-permutation? test with one-hot
        A1 = torch.zeros(M, K, device='cuda'); A1[:, 0] = 1.0
        W1 = torch.zeros(N, K, device='cuda'); W1[:, 0] = torch.arange(N, device='cuda').float() + 1
        o = run(A1, W1); print(' onehot k0 row0:', o[0, :8].tolist(), ' col0:', o[:8, 0].tolist())
        for kk in [1, 2, 3, 4, 5, 8, 16, 31]:
            A1 = torch.zeros(M, K, device='cuda'); A1[:, kk] = 1.0
            W1 = torch.zeros(N, K, device='cuda'); W1[:, kk] = 2.0
            o = run(A1, W1); print(f' k={kk}: out[0,0]={o[0,0].item()} (want 2) mean={o.mean().item()}')
        break
