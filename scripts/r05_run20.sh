#!/bin/bash
# round-5 GPU call 20: the whole GPU suite and smoke() on the final commit (printed in full)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
( time timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -12 ) > $O/r05_v_pytest_gpu.txt 2>&1; tail -8 $O/r05_v_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
