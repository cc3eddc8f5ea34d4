#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
for ts in 1 3 2; do
  DSH_GEMM_TILE=$ts timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-chain-latency > $O/x_bench_$ts.json 2> $O/x.err
  echo "DSH_GEMM_TILE=$ts"; python scripts/bench_brief.py $O/x_bench_$ts.json | grep "frames/s\|gemm_nt"
done
