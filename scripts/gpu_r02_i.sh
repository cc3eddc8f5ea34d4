#!/bin/bash
# fp32 GEMM tile-shape selection: parity tests, then config-2 bench per forced shape; chain trace with per-kernel listing
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_eval.py -m gpu -x -q 2>&1 | tail -4
for ts in 0 1 2 3; do
  DSH_GEMM_TILE=$ts timeout 200 python bench.py --dataset beat --precision fp32 --batch 256 --steps 2 --warmup 1 --no-cpu-baseline --no-chain-latency > $O/i_fp32_tile$ts.json 2> $O/i_fp32_tile$ts.err
  python - <<PY
import json
try:
    d = json.load(open("$O/i_fp32_tile$ts.json")); r = d["roofline"]
    print("DSH_GEMM_TILE=$ts", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 1), "ms/step; gemm", round(r["achieved"], 1), "TF/s frac", round(r["frac"], 3))
except Exception as e: print("tile $ts ERR", e)
PY
done
bash scripts/prof_chain.sh i 1 2>&1 | grep -v simple_timer | head -8
