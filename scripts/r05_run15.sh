#!/bin/bash
# round-5 GPU call 15: q|k|v as four-wave blocks (DSH_TL2_W4), wave priorities inside a SIMD (DSH_TL2_PRIO)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
echo "== bit identity"; timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "four_wave or rolling_main" 2>&1 | tail -5
echo "== q|k|v alone"
for cfg in "DSH_TL2_W4=0" "DSH_TL2_W4=1" "DSH_TL2_PRIO=1" "DSH_TL2_W4=0" "DSH_TL2_W4=1"; do
  echo "-- $cfg" | tee -a $O/r05_q_qkv_w4_timings.txt
  env $cfg timeout 200 python scripts/bench_tl2.py qkv 2>&1 | grep -v amdgpu.ids | grep "gen2\|span" | tee -a $O/r05_q_qkv_w4_timings.txt
done
echo "== bench A/B"
for cfg in "DSH_TL2_W4=1" "DSH_TL2_W4=0" "DSH_TL2_PRIO=1" "DSH_TL2_W4=1" "DSH_TL2_W4=0" "DSH_TL2_PRIO=1"; do
  env $cfg timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-16s %9.1f frames/s  %7.2f ms/step' % ('$cfg', d['value'], d['ms_per_step']))" | tee -a $O/r05_q_ab_qkv_w4.txt
done
