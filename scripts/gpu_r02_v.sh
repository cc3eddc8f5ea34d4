#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for ng in 0 1; do
  if [ $ng = 1 ]; then export DSH_NO_GRAPH=1; else unset DSH_NO_GRAPH; fi
  echo "NO_GRAPH=$ng"
  timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['chain_window_latency']; print({k:round(v,2) for k,v in c['chains_1'].items()}); print({k:round(v,2) for k,v in c['chains_16'].items()})"
done
