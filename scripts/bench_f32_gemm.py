"""Micro-benchmark of the fp32 parity path's Linears at configs[1]'s shape (BEAT, 256 clips x 34 frames = 8704 token rows):
gemm_nt_kernel<float> alone, and the round-6 launches that carry their LayerNorm / StylizationBlock front (gemm_f32_pro.hip)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsheg_amd import _lib
L = _lib.lib()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8704
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


rows = []
for name, n, k, act, res in [("qkv", 1536, 512, 0, False), ("sty.out", 512, 512, 0, True), ("feat_proj.1", 1024, 960, 1, False),
                             ("feat_proj.3", 512, 1024, 0, True), ("ffn.linear1", 1024, 512, 2, False), ("ffn.linear2", 512, 1024, 0, False)]:
    A, W, b = rn(M, k), rn(n, k) / k ** 0.5, rn(n)
    R = rn(M, n) if res else None
    Cf = R if res else torch.empty(M, n, device=dev)
    us = timeit(lambda: _lib.check(L.dsh_op_gemm(None, 0, P(A), P(W), P(b), P(R), P(Cf), None, M, n, k, act)))
    rows.append((f"gemm_nt {name}", n, k, us))
    us = timeit(lambda: _lib.check(L.dsh_op_gemm_f32_pro(None, 0, P(A), k, k, None, 0, 0, None, 0, 0, None, 0, 0, k, P(W), P(b), None, None, 0, 0, 1, 1, P(R), P(Cf), M, n, act, None, 0, None)))
    rows.append((f"pro0    {name}", n, k, us))
# folded LayerNorm launches
x = rn(M, 512); W = rn(1536, 512) / 23; b, fc = rn(1536), rn(1536); out = torch.empty(M, 1536, device=dev)
us = timeit(lambda: _lib.check(L.dsh_op_gemm_f32_pro(None, 1, P(x), 512, 512, None, 0, 0, None, 0, 0, None, 0, 0, 512, P(W), P(b), P(fc), None, 0, 0, 1, 1, None, P(out), M, 1536, 0, None, 0, None)))
rows.append(("pro1 LN+qkv", 1536, 512, us))
s1, s2, s3 = rn(M, 256), rn(M, 128), rn(M, 64); W = rn(1024, 960) / 31; b, fc = rn(1024), rn(1024); out = torch.empty(M, 1024, device=dev)
us = timeit(lambda: _lib.check(L.dsh_op_gemm_f32_pro(None, 1, P(x), 512, 512, P(s1), 256, 256, P(s2), 128, 128, P(s3), 64, 64, 947, P(W), P(b), P(fc), None, 0, 0, 1, 1, None, P(out), M, 1024, 1, None, 0, None)))
rows.append(("pro1 concat-LN+feat_proj.1", 1024, 960, us))
xx = rn(M, 960)
us = timeit(lambda: _lib.check(L.dsh_op_gemm_f32_pro(None, 1, P(xx), 960, 960, None, 0, 0, None, 0, 0, None, 0, 0, 947, P(W), P(b), P(fc), None, 0, 0, 1, 1, None, P(out), M, 1024, 1, None, 0, None)))
rows.append(("pro1 one-segment K=960", 1024, 960, us))
# StylizationBlock launch
nb = max(1, M // 34); film = rn(nb, 1024); W = rn(512, 512) / 23; b = rn(512); h = rn(M, 512)
us = timeit(lambda: _lib.check(L.dsh_op_gemm_f32_pro(None, 2, P(x), 512, 512, None, 0, 0, None, 0, 0, None, 0, 0, 512, P(W), P(b), None, P(film), 1024, 0, 34, nb, P(h), P(h), M, 512, 0, None, 0, None)))
rows.append(("pro2 StylizationBlock", 512, 512, us))
st = torch.zeros(M, 16, 2, device=dev); st[..., 1] = 32.0
us = timeit(lambda: _lib.check(L.dsh_op_gemm_f32_pro(None, 2, P(x), 512, 512, None, 0, 0, None, 0, 0, None, 0, 0, 512, P(W), P(b), None, P(film), 1024, 0, 34, nb, P(h), P(h), M, 512, 0, P(st), 16, None)))
rows.append(("pro2 with the producer's moments", 512, 512, us))
A2, W2, b2 = rn(M, 1024), rn(512, 1024) / 32, rn(512); y2 = torch.empty(M, 512, device=dev)
us = timeit(lambda: _lib.check(L.dsh_op_gemm_f32_pro(None, 0, P(A2), 1024, 1024, None, 0, 0, None, 0, 0, None, 0, 0, 1024, P(W2), P(b2), None, None, 0, 0, 1, 1, None, P(y2), M, 512, 0, None, 0, P(st))))
rows.append(("pro0 ffn.linear2 + moments out", 512, 1024, us))
for name, n, k, us in rows:
    fl = 2.0 * M * n * k
    print(f"{name:30s} M={M} N={n:5d} K={k:5d}: {us:7.1f} us  {fl / us / 1e6:6.1f} TF/s ({fl / us / 1e6 / 157.3 * 100:4.1f} % of 157.3)")

# linear attention core of configs[1]: 256 clips x 34 frames, 8 heads of 64 channels (fp32)
nbq = max(1, M // 34); qkv = rn(nbq, 34, 1536) * 2; yo = torch.empty(nbq, 34, 512, device=dev)
us = timeit(lambda: _lib.check(L.dsh_op_linear_attention(None, P(qkv), nbq, 34, 512, 64, P(yo))))
print(f"{'linear attention core (fp32)':30s} nb={nbq} T=34 D=512: {us:7.1f} us  ({(qkv.numel() + yo.numel()) * 4 / us / 1e6:6.2f} TB/s of q|k|v + y)")
