"""Micro-benchmark of the LDS-tiled kernel class (tl4.hip) against the register-stationary kernels (tl2.hip) at the bench shapes:
feat_proj.1 (concat + folded LayerNorm + SiLU; exp K = 896, ges K = 999 of 1024), feat_proj.3 on hi / lo residual planes, q|k|v.
Timing runs use DSH_TL_RAW=1 (operands passed through as if already tiled / permuted: results are garbage, the work is identical).
Usage: python scripts/bench_tl4.py [rows_per_half]   (default 83600 = 950 clips x 88 frames)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsheg_amd import _lib
L = _lib.lib(); P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
dev = "cuda"

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

Mc = int(sys.argv[1]) if len(sys.argv) > 1 else 83600
os.environ["DSH_TL_RAW"] = "1"; os.environ["DSH_TL2"] = "1"; os.environ["DSH_TL4_MIN_ROWS"] = "0"
# name, rows, K, N, pro, act, residual planes, kreal
cases = [("sa proj_out", 2 * Mc, 512, 512, 2, 0, True, 512), ("feat_proj.1 exp", Mc, 1024, 1024, 3, 1, False, 896), ("feat_proj.1 ges", Mc, 1024, 1024, 3, 1, False, 999),
         ("feat_proj.3", Mc, 1024, 512, 0, 0, True, 1024), ("q|k|v", 2 * Mc, 512, 1536, 1, 0, False, 512)]
for name, Mv, K, n, pro, act, res, kreal in cases:
    M = (Mv + 255) // 256 * 256 + 256
    torch.manual_seed(0)
    X = (torch.randn(M, K, device=dev) * 1.5 + 0.3).bfloat16(); W = (torch.randn(n, K, device=dev) / K ** 0.5).bfloat16()
    b = torch.randn(n, device=dev)
    gam = 1 + 0.1 * torch.randn(K, device=dev); bet = 0.1 * torch.randn(K, device=dev)
    # raw mode + residual planes: R = hi plane (bf16 bits in a float buffer of the right size is not needed: the op's raw mode takes R as is)
    R = torch.randn(M, n // 2, device=dev) if res else None          # [M, n] bf16 worth of bytes
    film = 0.3 * torch.randn(1900, 2 * K, device=dev) if pro == 2 else None
    Rlo = None
    Ct = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    Cf = torch.empty(M, n, device=dev) if res else None
    os.environ["DSH_HILO"] = "1" if res else "0"
    def run():
        _lib.check(L.dsh_op_tl_linear(None, pro, P(X), P(W), P(b), P(R), P(Cf), P(Ct), Mv, n, act, P(gam), P(bet), P(film), kreal if pro == 3 else 88, 1900 if pro == 2 else 1, K))
    fl_pad = 2.0 * Mv * n * K; fl = 2.0 * Mv * n * kreal
    line = [f"{name:16s} M={Mv:6d} K={kreal:4d} N={n:4d}"]
    for tag, env in (("tl2", {"DSH_TL4": "0", "DSH_TL2_ROT": "0", "DSH_TL2_KSKIP": "0"}), ("tl2kskip", {"DSH_TL4": "0", "DSH_TL2_ROT": "0", "DSH_TL2_KSKIP": "1"}), ("tl4a", {"DSH_TL4": "7", "DSH_TL4_V": "a"}), ("tl4b", {"DSH_TL4": "7", "DSH_TL4_V": "b"})):
        os.environ.update(env)
        try:
            us = timeit(run)
            fam = L.dsh_debug_last_tl_variant()
            line.append(f"{tag}[{fam}] {us:7.1f} us {fl/us/1e6:6.0f} TF/s")
        except Exception as e:
            line.append(f"{tag} failed: {str(e)[:80]}")
    print(" | ".join(line), flush=True)
    if pro == 3 and kreal == 999 and os.environ.get("TL4_ABL", "1") != "0":
        for v in ("a", "b"):
            os.environ.update({"DSH_TL4": "7", "DSH_TL4_V": v})
            out = []
            for ab, what in ((1, "no DMA in loop"), (2, "no MFMA"), (3, "no moments"), (4, "no LDS reads")):
                os.environ["DSH_TL4_ABL"] = str(ab)
                out.append(f"{what} {timeit(run):6.1f} us")
            os.environ["DSH_TL4_ABL"] = "0"
            print(f"    ablations tl4{v}: " + " | ".join(out), flush=True)
