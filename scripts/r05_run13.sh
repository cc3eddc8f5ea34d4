#!/bin/bash
# round-5 GPU call 13: variants of the fused FFN's last stage (DSH_FFN_PB: 1 = hi plane of tiles 6..15 kept in registers, 2 = residual requested two phases ahead, 3 = both)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
echo "== bit identity"; timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "pipelined_phase_c" 2>&1 | tail -5
echo "== block timeline"
for pb in 0 1 2 3; do
  echo "-- DSH_FFN_PB=$pb (input = hi plane)" | tee -a $O/r05_o_ffn_pb_timeline.txt
  BENCH_FFN_VERS=3 DSH_HILO=1 DSH_FFN_X_IS_HI=1 DSH_FFN_PB=$pb timeout 200 python scripts/bench_tl2.py ffn 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O/r05_o_ffn_pb_timeline.txt
done
echo "== bench A/B"
for cfg in "DSH_FFN_PB=3" "DSH_FFN_PB=0" "DSH_FFN_PB=2" "DSH_FFN_PB=1" "DSH_FFN_PB=0" "DSH_FFN_PB=3" "DSH_FFN_PB=2"; do
  env $cfg timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-16s %9.1f frames/s  %7.2f ms/step' % ('$cfg', d['value'], d['ms_per_step']))" | tee -a $O/r05_o_ab_ffn_pb.txt
done
