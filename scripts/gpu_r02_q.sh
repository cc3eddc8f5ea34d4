#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for G in 32 16; do
  echo "G=$G default"; timeout 120 python scripts/run_chain_window.py $G 4 2>&1 | tail -1
  echo "G=$G no cache"; DSH_LEVEL_CACHE=0 timeout 120 python scripts/run_chain_window.py $G 4 2>&1 | tail -1
  echo "G=$G two streams (no cache)"; DSH_DUAL_MIN_ROWS=512 timeout 120 python scripts/run_chain_window.py $G 4 2>&1 | tail -1
  echo "G=$G four streams (no cache)"; DSH_DUAL=4 DSH_DUAL_MIN_ROWS=512 timeout 120 python scripts/run_chain_window.py $G 4 2>&1 | tail -1
done
