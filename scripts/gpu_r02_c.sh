#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r02_c}
timeout 300 python scripts/bench_tl2.py > gpurun_out/${TAG}_tl2.log 2>&1; echo "bench_tl2 rc=$?"; cat gpurun_out/${TAG}_tl2.log | tail -60
bash scripts/prof_chain.sh $TAG 1
