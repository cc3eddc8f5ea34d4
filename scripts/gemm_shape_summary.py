"""rocprofv3 --kernel-trace (rocpd sqlite) -> launches grouped by (kernel, grid size, LDS size): where a multi-shape kernel
(gemm_nt_kernel) spends its time.  usage: gemm_shape_summary.py <results.db> [name substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else "gemm_nt"
agg = {}
for n, g, lds, s, e in cur.execute("select name, grid_x, lds_size, start, end from kernels"):
    if pat not in n:
        continue
    a = agg.setdefault((n.replace("void dsh::", "")[:60], g, lds), [0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"{'kernel':60s} {'grid_x':>9s} {'lds':>7s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>8s} {'pct':>6s}")
for (n, g, lds), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:60s} {g:9d} {lds:7d} {c:6d} {t/1e3:9.2f} {t/c:8.1f} {100*t/tot:6.1f}")
