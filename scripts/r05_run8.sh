#!/bin/bash
# round-5 GPU call 8: sub-batch boundaries that make the 256-token-block launches whole rounds (372 clips = 65 536 token rows)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
for cfg in "X=1" "DSH_SPLIT_AT=372,744" "DSH_SPLIT_AT=372,661" "DSH_SPLIT_AT=289,578" "X=1" "DSH_DUAL=4 DSH_SPLIT_AT=186,372,744" "DSH_DUAL=4 DSH_SPLIT_AT=372,558,744" "DSH_DUAL=4"; do
  env $cfg timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s %9.1f frames/s  %7.2f ms/step' % ('$cfg', d['value'], d['ms_per_step']))" | tee -a $O/r05_h_ab_split.txt
done
