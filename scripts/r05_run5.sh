#!/bin/bash
# round-5 GPU call 5: stores far in front of the counted waits — FFN pass-B epilogues in the first half of their phase (DSH_FFN_PC=2) and
# the rolling Linears' epilogue stores behind the mid-phase wait
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
echo "== op tests"; timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "ffn or rolling or pipelined or tl_linear" 2>&1 | tail -3
echo "== FFN timelines"
for pc in 1 2 1 2; do
  DSH_FFN_PC=$pc BENCH_FFN_VERS=3 DSH_HILO=1 timeout 200 python scripts/bench_tl2.py ffn 2>&1 | grep -v amdgpu.ids | sed "s/^/PC=$pc /" | tee -a $O/r05_e_ffn_block_timeline.txt
done
echo "== q|k|v in isolation"
for r in 0 1 1; do
  DSH_TL2_ROLL=$r timeout 300 python scripts/bench_tl2.py qkv 2>&1 | grep -v amdgpu.ids | grep "gen2\|span" | sed "s/^/ROLL=$r /" | tee -a $O/r05_e_qkv_roll_timings.txt
done
echo "== bench A/B"
for pc in 2 1 2 1; do
  DSH_FFN_PC=$pc timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PC=$pc %9.1f frames/s  %7.2f ms/step  enqueue %.1f launch-cost %.1f ms' % (d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d.get('host_launch_cost_ms_per_step', -1)))" | tee -a $O/r05_e_ab.txt
done
echo "== instrumented step"
timeout 300 python bench.py --no-cpu-baseline --no-chain-latency > $O/r05_e_bench.json 2> $O/r05_e_bench.err; python scripts/bench_brief.py $O/r05_e_bench.json
