#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_eval.py -m gpu -x -q 2>&1 | tail -2
for ts in 1 3 4; do
  DSH_GEMM_TILE=$ts timeout 200 python bench.py --dataset beat --precision fp32 --batch 256 --steps 2 --warmup 1 --no-cpu-baseline --no-chain-latency > $O/o_fp32_tile$ts.json 2> $O/o.err
  python - <<PY
import json
try:
    d = json.load(open("$O/o_fp32_tile$ts.json")); r = d["roofline"]
    print("DSH_GEMM_TILE=$ts", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 1), "ms/step; gemm", round(r["achieved"], 1), "TF/s frac", round(r["frac"], 3))
except Exception as e: print("tile $ts ERR", e)
PY
done
timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-chain-latency --no-roofline > $O/o_bench.json 2>> $O/o.err; python scripts/bench_brief.py $O/o_bench.json | head -2
for G in 1 16; do timeout 120 python scripts/run_chain_window.py $G 4 2>&1 | tail -1; done
