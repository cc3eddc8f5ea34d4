#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_eval.py -m gpu -x -q 2>&1 | tail -2
timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-chain-latency > $O/r_bench.json 2> $O/r.err; python scripts/bench_brief.py $O/r_bench.json | grep -v "^  tl\|^  gemm"
for G in 1 16; do timeout 120 python scripts/run_chain_window.py $G 4 2>&1 | tail -1; done
