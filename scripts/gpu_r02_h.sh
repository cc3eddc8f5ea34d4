#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r02_h}
timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|rror|FAILED" gpurun_out/${TAG}_pytest.log | tail -5
bash scripts/prof_chain.sh $TAG 1 2>&1 | grep -v "simple_timer"
