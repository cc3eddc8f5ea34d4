import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsheg_amd import _lib
L = _lib.lib(); P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
dev = "cuda"; K = 1024; Mv = 1280; M = 1280; n = 1024
torch.manual_seed(0)
X = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(n, K, device=dev) / K ** 0.5).bfloat16(); b = torch.randn(n, device=dev)
for act in [0, 1]:
    Ct = torch.zeros(M, n, device=dev, dtype=torch.bfloat16)
    _lib.check(L.dsh_op_tl_linear(None, 0, P(X), P(W), P(b), None, None, P(Ct), Mv, n, act, None, None, None, 88, 1, K))
    torch.cuda.synchronize()
    ref = X.float() @ W.float().T + b
    if act: ref = torch.nn.functional.silu(ref)
    got = Ct.float()
    bad = ~torch.isfinite(got)
    print("act", act, "nan count", int(bad.sum()), "max err (finite)", (got - ref)[~bad].abs().max().item())
    if bad.any():
        idx = bad.nonzero()[:10]; print(idx.tolist()); print("ref at bad:", ref[bad][:10].tolist())
for KK in [512, 1024]:
    X = torch.randn(M, KK, device=dev).bfloat16(); W = (torch.randn(n, KK, device=dev) / KK ** 0.5).bfloat16()
    Ct = torch.zeros(M, n, device=dev, dtype=torch.bfloat16)
    _lib.check(L.dsh_op_tl_linear(None, 0, P(X), P(W), P(b), None, None, P(Ct), Mv, n, 1, None, None, None, 88, 1, KK))
    torch.cuda.synchronize()
    lin = X.float() @ W.float().T + b
    got = Ct.float()
    for nm, r in [("silu", torch.nn.functional.silu(lin)), ("none", lin), ("gelu", torch.nn.functional.gelu(lin))]:
        print(KK, nm, (got - r).abs().max().item())
    e = (got - torch.nn.functional.silu(lin)).abs(); i = e.argmax(); print("worst at", (i // n).item(), (i % n).item(), "lin", lin.flatten()[i].item(), "got", got.flatten()[i].item())
