#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_eval.py -m gpu -x -q 2>&1 | tail -2
for mr in 32768 4096; do
  DSH_DUAL_MIN_ROWS=$mr timeout 200 python bench.py --dataset beat --precision fp32 --batch 256 --steps 2 --warmup 1 --no-cpu-baseline --no-chain-latency > $O/l_fp32_mr$mr.json 2> $O/l_fp32_mr$mr.err
  python - <<PY
import json
try:
    d = json.load(open("$O/l_fp32_mr$mr.json")); r = d["roofline"]
    print("DSH_DUAL_MIN_ROWS=$mr", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 1), "ms/step; gemm", round(r["achieved"], 1), "TF/s", d["config"].get("streams_per_gpu"))
except Exception as e: print("ERR", e)
PY
done
for G in 1 16; do timeout 120 python scripts/run_chain_window.py $G 4 2>&1 | tail -1; done
