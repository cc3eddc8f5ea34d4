#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r02_f}
timeout 300 python -m pytest tests/test_gpu_ops.py -q -s -k "tl_linear or tl2" > gpurun_out/${TAG}_ops.log 2>&1; echo "ops rc=$?"
grep -E "tl2_ffn|passed|failed|rror|FAILED" gpurun_out/${TAG}_ops.log | tail -8
timeout 300 python scripts/bench_tl2.py > gpurun_out/${TAG}_tl2.log 2>&1; echo "bench_tl2 rc=$?"; cat gpurun_out/${TAG}_tl2.log | grep -v amdgpu.ids | tail -40
timeout 300 python -m pytest tests/test_gpu_eval.py -q -s -x > gpurun_out/${TAG}_eval.log 2>&1; echo "eval rc=$?"
grep -o "\[[a-z0-9_ ]*[^]]*\][^\[]*" gpurun_out/${TAG}_eval.log | grep -E "err|max|rel" | head -12; grep -E "passed|failed|rror" gpurun_out/${TAG}_eval.log | tail -3
for v in "1 1" "1 0"; do set -- $v
  DSH_TL2=$1 DSH_FFN_FUSE=$2 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-chain-latency > gpurun_out/${TAG}_bench_$1$2.json 2> gpurun_out/${TAG}_bench_$1$2.err; echo "bench tl2=$1 fuse=$2 rc=$?"
  python scripts/bench_brief.py gpurun_out/${TAG}_bench_$1$2.json 2>&1 | tail -12
done
