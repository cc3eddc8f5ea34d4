#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
echo "== op tests"; timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "rolling or pipelined or tl_linear or hilo" 2>&1 | tail -15
echo "== full gpu suite"; timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -12 | tee $O/r05_g_pytest_gpu.txt
