"""rocprofv3 --kernel-trace (rocpd sqlite) of scripts/run_chain_window.py -> per-kernel table + launch-gap accounting.
usage: chain_trace_summary.py <results.db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
# the last window = the last third of the trace (3 identical windows): take kernels after the last big idle gap (> 2 ms)
cut = 0
for i in range(1, len(rows)):
    if rows[i][1] - rows[i - 1][2] > 2_000_000: cut = i
rows = rows[cut:]
span = (rows[-1][2] - rows[0][1]) / 1e3
busy = sum(r[2] - r[1] for r in rows) / 1e3
gaps = [max(0, rows[i][1] - rows[i - 1][2]) / 1e3 for i in range(1, len(rows))]
print(f"# last window: {len(rows)} kernels, span {span/1e3:.2f} ms, kernel time {busy/1e3:.2f} ms ({100*busy/span:.0f} %), "
      f"gaps {sum(gaps)/1e3:.2f} ms (mean {sum(gaps)/len(gaps):.2f} us, median {sorted(gaps)[len(gaps)//2]:.2f} us)")
agg = {}
for n, s, e in rows:
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
print(f"{'kernel':90s} {'calls':>6s} {'total_us':>10s} {'avg_us':>8s} {'pct':>6s}")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n[:90]:90s} {c:6d} {t:10.1f} {t/c:8.2f} {100*t/busy:6.2f}")

# ---- the last evaluation in launch order (name, grid size when the view exposes it, duration, gap to the previous kernel)
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
gcol = next((c for c in ("grid_x", "grid_size_x", "grid_size", "workgroup_x") if c in cols), None)
q = f"select name, start, end{', ' + gcol if gcol else ''} from kernels order by start"
allr = list(cur.execute(q))[-200:]
print(f"\n# last {len(allr)} kernels in launch order (columns of the kernels view: {', '.join(cols)})")
for i, r in enumerate(allr):
    gap = (r[1] - allr[i - 1][2]) / 1e3 if i else 0.0
    short = r[0].replace("void dsh::", "").replace("dsh::", "")[:70]
    print(f"{short:70s} grid {str(r[3]) if gcol else '?':>8s}  {(r[2] - r[1]) / 1e3:7.2f} us  gap {gap:6.2f}")
