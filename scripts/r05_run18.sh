#!/bin/bash
# round-5 GPU call 18: start lag between the sub-batch streams (DSH_DUAL_LAG, launches; default 3) and stream count on the final build
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
for cfg in "DSH_DUAL_LAG=3" "DSH_DUAL_LAG=0" "DSH_DUAL_LAG=8" "DSH_DUAL_LAG=24" "DSH_DUAL_LAG=60" "DSH_DUAL_LAG=3" "DSH_DUAL=4" "DSH_DUAL=5" "DSH_DUAL_LAG=24" "DSH_DUAL_LAG=0"; do
  env $cfg timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); t=d.get('telemetry') or {}
print('%-18s %9.1f frames/s  %7.2f ms/step  sclk %6.0f MHz %6.0f W' % ('$cfg', d['value'], d['ms_per_step'], t.get('clock_mhz_mean',0), t.get('power_w_mean',0)))" | tee -a $O/r05_t_ab_lag_streams.txt
done
