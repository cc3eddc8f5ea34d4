#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sampler.py -m gpu -x -q 2>&1 | tail -2
timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['chain_window_latency']['chains_1'], d['chain_window_latency']['chains_16'])"
timeout 200 python bench.py --mode chain --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('chain mode', round(d['value'],1), round(d['ms_per_step'],1))"
