"""Micro-benchmark of the token-per-lane kernels at the bench shapes (M = 167 200 / 83 600 rows): first generation
(tl_linear.hip, register-staged W) vs second generation (tl2.hip, LDS-DMA from fragment-ordered W), the fused FFN kernel,
and the per-block timeline of each (DSH_TL_TRACE).  Timing runs use DSH_TL_RAW=1 (operands passed through as if already
tiled / permuted: results are garbage, the work is identical)."""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsheg_amd import _lib
L = _lib.lib(); P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
dev = "cuda"
os.makedirs("gpurun_out", exist_ok=True)

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

def trace_summary(path):
    rows = [[int(v) for v in l.split()] for l in open(path)]
    rows = [r for r in rows if r[1]]
    t0 = min(r[1] for r in rows)
    span = (max(r[3] for r in rows) - t0) / 100.0
    pro = [(r[2] - r[1]) / 100.0 for r in rows]; dur = [(r[3] - r[1]) / 100.0 for r in rows]
    starts = sorted((r[1] - t0) / 100.0 for r in rows)
    # blocks that started in the first 2 us = first round; count of rounds ~ blocks / first-round blocks
    first = sum(1 for s in starts if s < 2.0)
    return (f"span {span:7.1f} us | blocks {len(rows)} first-round {first} | block dur med {statistics.median(dur):6.1f} max {max(dur):6.1f} us | "
            f"prologue med {statistics.median(pro):5.1f} max {max(pro):5.1f} us")

def probe_summary(path):
    """per-iteration phase cycles (wave 0 of every block): VMEM issue | MFMA phase | vmcnt wait | barrier wait; clock"""
    rows = [[int(v) for v in l.split()] for l in open(path)]
    rows = [r for r in rows if r[5] > 1]
    med = lambda xs: statistics.median(xs)
    ph = [med([r[1 + k] / (r[5] - 1) for r in rows]) for k in range(4)]
    clk = med([r[6] / (r[7] / 100.0) / 1e3 for r in rows if r[7]])       # cycles per us / 1e3 = GHz
    tot = sum(ph)
    return (f"per phase of 32 MFMAs (median over {len(rows)} blocks, cycles): setup {ph[0]:7.0f} | mfma+vmem groups {ph[1]:7.0f} | epilogue {ph[2]:7.0f} | "
            f"wait+barrier {ph[3]:7.0f} | total {tot:7.0f}; shader clock {clk:.2f} GHz")

T, nb = 88, 950
cases = [("qkv", 167200, 512, 1536, 1, 0, False, False, True), ("sty", 167200, 512, 512, 2, 0, True, True, True),
         ("ffn1", 167200, 512, 1024, 0, 2, False, False, True), ("ffn2", 167200, 1024, 512, 0, 0, False, False, True),
         ("feat3", 83600, 1024, 512, 0, 0, True, True, True), ("feat1p0", 83600, 1024, 1024, 0, 1, False, False, True)]
only = sys.argv[1].split(",") if len(sys.argv) > 1 else None
os.environ["DSH_TL_RAW"] = "1"
for name, Mv, K, n, pro, act, res, cf, ct in cases:
    if only and name not in only: continue
    M = (Mv + 255) // 256 * 256 + 256
    torch.manual_seed(0)
    X = (torch.randn(M, K, device=dev) * 1.5 + 0.3).bfloat16(); W = (torch.randn(n, K, device=dev) / K ** 0.5).bfloat16()
    b = torch.randn(n, device=dev); R = torch.randn(M, n, device=dev) if res else None
    gam = 1 + 0.1 * torch.randn(K, device=dev); bet = 0.1 * torch.randn(K, device=dev)
    film = 0.3 * torch.randn(nb * 2, 2 * K, device=dev)
    Cf = torch.empty(M, n, device=dev) if cf else None; Ct = torch.empty(M, n, device=dev, dtype=torch.bfloat16) if ct else None
    def run():
        _lib.check(L.dsh_op_tl_linear(None, pro, P(X), P(W), P(b), P(R), P(Cf), P(Ct), Mv, n, act, P(gam), P(bet), P(film), T, nb * 2, K))
    fl = 2.0 * Mv * n * K
    out = []
    for gen in ("0", "1"):
        os.environ["DSH_TL2"] = gen
        os.environ.pop("DSH_TL_TRACE", None)
        us = timeit(run)
        out.append(f"gen{int(gen)+1} {us:7.1f} us {fl/us/1e6:7.1f} TF/s")
        if gen == "1":
            tp = f"gpurun_out/trace_{name}_gen2.txt"
            os.environ["DSH_TL_TRACE"] = tp
            run(); torch.cuda.synchronize()
            os.environ.pop("DSH_TL_TRACE", None)
            try: out.append("   [" + trace_summary(tp) + "]")
            except Exception as e: out.append(f"   [trace: {e}]")
        if gen == "1" and name in ("qkv", "sty", "ffn2"):
            pbf = f"gpurun_out/probe_{name}.txt"
            os.environ["DSH_TL_PROBE"] = pbf
            run(); torch.cuda.synchronize()
            os.environ.pop("DSH_TL_PROBE", None)
            try: out.append("   [" + probe_summary(pbf) + "]")
            except Exception as e: out.append(f"   [probe: {e}]")
            for dbg, what in ((8, "W L2-hot"),):
                os.environ["DSH_TL_DBG"] = str(dbg)
                out.append(f"   ablation {what:18s}: {timeit(run):7.1f} us")
            os.environ.pop("DSH_TL_DBG", None)
    print(f"TL {name:8s} M={Mv} K={K} N={n} pro={pro}:\n   " + "\n   ".join(out), flush=True)
    del X, W, R, Cf, Ct
os.environ.pop("DSH_TL_RAW", None)

# fused FFN (one launch instead of ffn1 + ffn2 + sty): timed through the block timeline of the last of 3 launches; both kernel
# generations (DSH_FFN_V = 2: tl2_ffn_kernel, 3: tl3_ffn_kernel) with their phase probes
if not only or "ffn" in only:
    Mv = 167200; M = (Mv + 127) // 128 * 128
    D, F = 512, 1024
    torch.manual_seed(1)
    X = (torch.randn(M, D, device=dev) * 1.2).bfloat16(); H = torch.randn(M, D, device=dev)
    W1 = (torch.randn(F, D, device=dev) / D ** 0.5).bfloat16(); W2 = (torch.randn(D, F, device=dev) / F ** 0.5).bfloat16()
    W3 = (torch.randn(D, D, device=dev) / D ** 0.5).bfloat16()
    b1, b2, b3 = torch.randn(F, device=dev), torch.randn(D, device=dev), torch.randn(D, device=dev)
    gam, bet = 1 + 0.1 * torch.randn(D, device=dev), 0.1 * torch.randn(D, device=dev)
    film = 0.3 * torch.randn(nb * 2, 2 * D, device=dev)
    Cf = torch.empty(M, D, device=dev); Ct = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    def ffn():
        _lib.check(L.dsh_op_tl2_ffn(None, P(X), P(H), P(W1), P(b1), P(W2), P(b2), P(W3), P(b3), P(gam), P(bet), P(film), T, nb * 2, None, 0,
                                    P(Cf), P(Ct), Mv))
        torch.cuda.synchronize()
    fl = 2.0 * Mv * (2 * D * F + D * D)
    for ver in (os.environ.get("BENCH_FFN_VERS", "2,3").split(",")):
        os.environ["DSH_FFN_V"] = ver
        os.environ["DSH_FFN_REPEAT"] = "2"
        tp = f"gpurun_out/trace_ffn_v{ver}.txt"
        os.environ["DSH_TL_TRACE"] = tp
        ffn()
        rows = [[int(v) for v in l.split()] for l in open(tp)]
        span = (max(r[3] for r in rows) - min(r[1] for r in rows)) / 100.0
        print(f"fused FFN v{ver} M={Mv}: {span:7.1f} us  {fl/span/1e6:7.1f} TF/s   [" + trace_summary(tp) + "]")
        os.environ.pop("DSH_TL_TRACE", None); os.environ["DSH_FFN_REPEAT"] = "0"
        pf = f"gpurun_out/probe_ffn_v{ver}.txt"
        os.environ["DSH_TL_PROBE"] = pf
        ffn()
        os.environ.pop("DSH_TL_PROBE", None)
        rows = [[int(v) for v in l.split()] for l in open(pf)]
        med = statistics.median
        if ver == "2":
            print("   phase C [" + probe_summary(pf) + "]  (ideal: 1024 cycles per 32-MFMA phase)")
            rows = [r for r in rows if len(r) > 8 and r[8]]
            if rows:
                endC = med(r[6] for r in rows); endLN = med(r[8] & 0xffffffff for r in rows); end = med(r[8] >> 32 for r in rows)
                loopC = med(sum(r[1:5]) for r in rows)
                print(f"   block timeline (median, shader cycles since block start): prologue {endC - loopC:7.0f} | phase C {loopC:7.0f} | LayerNorm / FiLM / SiLU stage "
                      f"{endLN - endC:7.0f} | phase D + last epilogue {end - endLN:7.0f} | total {end:7.0f}")
        else:
            rows = [r for r in rows if len(r) > 7 and r[6]]
            if rows:
                st = [med(r[1 + k] for r in rows) for k in range(6)]
                clk = med(r[6] / (r[7] / 100.0) / 1e3 for r in rows if r[7])
                print(f"   block timeline v3 (median over {len(rows)} blocks, shader cycles): prologue {st[0]:7.0f} | phase C {st[1] - st[0]:7.0f} ({(st[1] - st[0]) / 64:5.0f} per phase) | "
                      f"row statistics + first conversion {st[2] - st[1]:7.0f} | pass A (4 phases) {st[3] - st[2]:7.0f} | pass B (12 phases) {st[4] - st[3]:7.0f} | "
                      f"last epilogue {st[5] - st[4]:7.0f} | total {st[5]:7.0f}; shader clock {clk:.2f} GHz"
                      + (f"; of pass B {med(r[8] for r in rows):7.0f} cycles at its 12 phase tops (counted wait + barrier)" if len(rows[0]) > 8 else ""))
    os.environ.pop("DSH_FFN_V", None)
