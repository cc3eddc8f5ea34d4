#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-chain-latency > $O/n_bench.json 2> $O/n.err; python scripts/bench_brief.py $O/n_bench.json
for G in 1 16; do timeout 120 python scripts/run_chain_window.py $G 4 2>&1 | tail -1; done
