#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
for mr in 32768 16384; do
  DSH_DUAL_MIN_ROWS=$mr timeout 200 python bench.py --mode ddpm --batch 313 --steps 1 --warmup 0 > $O/s_ddpm_$mr.json 2> $O/s.err
  python -c "
import json; d=json.load(open('$O/s_ddpm_$mr.json')); print('MIN_ROWS=$mr', round(d['value'],1), d['unit'], round(d['ms_per_step'],1), 'ms')"
done
