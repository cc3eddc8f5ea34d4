#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_eval.py -m gpu -x -q 2>&1 | tail -2
timeout 200 python bench.py --dataset beat --precision fp32 --batch 256 --steps 3 --warmup 1 --no-cpu-baseline --no-chain-latency > $O/z_fp32.json 2> $O/z.err
python -c "
import json; d=json.load(open('$O/z_fp32.json')); r=d['roofline']; print('fp32', round(d['value'],1), d['unit'], round(d['ms_per_step'],1), 'ms; gemm', round(r['achieved'],1), 'TF/s', round(r['frac'],3), 'attn ms/step', round(r.get('attention_ms_per_step',0),1))"
