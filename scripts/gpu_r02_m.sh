#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
for cfgs in "2 3" "3 3" "4 3" "2 6" "3 6"; do
  set -- $cfgs
  DSH_DUAL=$1 DSH_DUAL_LAG=$2 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-chain-latency --no-roofline > $O/m_dual$1_lag$2.json 2> $O/m.err
  python - <<PY
import json
try:
    d = json.load(open("$O/m_dual$1_lag$2.json")); print("DSH_DUAL=$1 LAG=$2", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 1), "ms/step")
except Exception as e: print("ERR", e)
PY
done
