#!/usr/bin/env python3
"""Summarise a rocprofv3 (--kernel-trace --stats, rocpd sqlite output) run into a text table:
per-kernel totals and, for the GEMM kernel, a per-launch-shape breakdown.
usage: rocprof_summary.py <results.db> [n_bench_steps_in_trace]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cur = db.cursor()
rows = list(cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                        "from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"# rocprofv3 --kernel-trace --stats summary; {steps:g} bench step(s) in trace; total kernel time {tot:.1f} ms")
print(f"{'kernel':78s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
for r in rows:
    print(f"{r[0][:78]:78s} {r[1]:7d} {r[2]:10.2f} {r[3]:9.1f} {r[4]:9.1f} {r[5]:9.1f} {100*r[2]/tot:6.2f}")
print()
print("# GEMM launches by grid (blocks_n x blocks_m of 128x128 tiles)")
for name in [r[0] for r in rows if "gemm" in r[0]]:
    print(name)
    q = ("select grid_x/workgroup_x, grid_y, count(*), sum(end-start)/1e6, avg(end-start)/1e3, vgpr_count, accum_vgpr_count, lds_size "
         "from kernels where name = ? group by grid_x, grid_y order by 4 desc")
    print(f"  {'blk_n':>6s} {'blk_m':>6s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} vgpr agpr lds")
    for g in cur.execute(q, (name,)):
        print(f"  {g[0]:6d} {g[1]:6d} {g[2]:7d} {g[3]:10.2f} {g[4]:9.1f} {g[5]} {g[6]} {g[7]}")
