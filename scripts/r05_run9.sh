#!/bin/bash
# round-5 GPU call 9: what each launch of a layer costs the 3-stream STEP (DSH_DBG_SKIP: the launch is simply not issued; results are garbage)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
for cfg in "DSH_DBG_SKIP=0" "DSH_DBG_SKIP=1" "DSH_DBG_SKIP=2" "DSH_DBG_SKIP=4" "DSH_DBG_SKIP=8" "DSH_DBG_SKIP=16" "DSH_DBG_SKIP=32" "DSH_DBG_SKIP=26" "DSH_DBG_SKIP=37" "DSH_DBG_SKIP=0"; do
  env $cfg timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-20s %9.1f frames/s  %7.2f ms/step' % ('$cfg', d['value'], d['ms_per_step']))" | tee -a $O/r05_i_marginal_cost.txt
done
