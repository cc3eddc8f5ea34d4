#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r02_g}
timeout 600 python -m pytest tests -m gpu -q -s > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|rror|FAILED" gpurun_out/${TAG}_pytest.log | tail -8
grep -o "\[[a-z0-9_ ]*[^]]*\][^\[]*" gpurun_out/${TAG}_pytest.log | grep -E "bf16|tl2" | head -20
for v in "2 1" "0 1" "2 0"; do set -- $v
  DSH_DUAL=$1 DSH_FFN_FUSE=$2 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-chain-latency --no-roofline > gpurun_out/${TAG}_bench_d$1f$2.json 2> gpurun_out/${TAG}_bench_d$1f$2.err; echo "bench dual=$1 fuse=$2 rc=$?"
  python -c "import json; d=json.load(open('gpurun_out/${TAG}_bench_d$1f$2.json')); print(round(d['value'],1), 'frames/s', round(d['ms_per_step'],1), 'ms/step')"
done
