#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_eval.py -m gpu -x -q 2>&1 | tail -2
timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['chain_window_latency']; print(round(d['value'],1)); print({k:round(v,2) for k,v in c['chains_1'].items()}); print({k:round(v,2) for k,v in c['chains_16'].items()})"
timeout 200 python bench.py --mode chain --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('chain mode', round(d['value'],1), round(d['ms_per_step'],1))"
