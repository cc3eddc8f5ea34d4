#!/bin/bash
# timestep cache + single-transformer variant: full GPU suite, then chain-window latency with / without the cache
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
for lc in 0 1; do for G in 1 16; do
  echo "DSH_LEVEL_CACHE=$lc chains=$G"; DSH_LEVEL_CACHE=$lc timeout 120 python scripts/run_chain_window.py $G 4 2>&1 | tail -2
done; done
