#!/bin/bash
# round-5 GPU call 14: evidence of the build with DSH_FFN_PB=1 as the default (profiles r05_p_*) + the whole GPU suite
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
bash scripts/gpu_profiles.sh r05_p 2>&1 | tail -60
echo "== GPU suite"
( time timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 ) > $O/r05_p_gpu_tests.txt 2>&1; tail -8 $O/r05_p_gpu_tests.txt
