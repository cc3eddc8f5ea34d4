#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sampler.py -m gpu -x -q 2>&1 | tail -3
for pf in 1 0; do for G in 1 16; do
  echo "DSH_LEVEL_PREFETCH=$pf chains=$G"; DSH_LEVEL_PREFETCH=$pf timeout 120 python scripts/run_chain_window.py $G 4 2>&1 | tail -1
done; done
timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['chain_window_latency']['chains_1'], d['chain_window_latency']['chains_16'])"
