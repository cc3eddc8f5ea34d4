#!/bin/bash
# kernel trace of one chained window at batch G -> gpurun_out/<tag>_chain_trace_G.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r02}; G=${2:-1}
rm -rf gpurun_out/prof_chain; mkdir -p gpurun_out/prof_chain
DSH_NO_GRAPH=${DSH_NO_GRAPH:-} timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_chain -o p -- python scripts/run_chain_window.py $G 3 > gpurun_out/prof_chain/run.log 2>&1
tail -3 gpurun_out/prof_chain/run.log
DB=$(find gpurun_out/prof_chain -name "*.db" | head -1)
python scripts/chain_trace_summary.py $DB > gpurun_out/${TAG}_chain_trace_$G.txt 2>&1; head -40 gpurun_out/${TAG}_chain_trace_$G.txt
rm -rf gpurun_out/prof_chain
