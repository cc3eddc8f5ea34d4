#!/bin/bash
# round-5 GPU call 4: remaining new tests; where the host's enqueue time goes (HIP API stats of one step); mid-size / fp32 baselines of the day
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_ops_golden.py tests/test_gpu_ops.py -q -m gpu -s -k "fixtures or rolling or pipelined or nonfinite" 2>&1 | grep -v amdgpu.ids | grep "ops_show\|passed\|failed\|Error\|error" | tee $O/r05_d_new_tests.txt
echo "== hip api stats"
D=$O/prof_r05_d_hip; rm -rf $D; mkdir -p $D
timeout 300 rocprofv3 --hip-trace --stats -d $D -o p --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-chain-latency > $D/bench.log 2>&1
F=$(find $D -name "*hip_api_stats*.csv" | head -1); echo "stats file: $F"; head -12 "$F" | cut -c1-200 | tee $O/r05_d_hip_api_stats.txt
tail -1 $D/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('under hip-trace: ms/step %.1f enqueue %.1f' % (d['ms_per_step'], d['host_enqueue_ms_per_step']))"
rm -rf $D
echo "== enqueue by stream count"
for dual in 0 3; do
  DSH_DUAL=$dual timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('DUAL=$dual %9.1f frames/s  %7.2f ms/step enqueue %.1f ms' % (d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step']))" | tee -a $O/r05_d_enqueue.txt
done
echo "== baselines of the day: ddpm313, 100 clips, config 2"
timeout 300 python bench.py --mode ddpm --batch 313 --steps 1 --warmup 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ddpm313 %9.1f frames/s  %7.2f ms/step' % (d['value'], d['ms_per_step']))" | tee -a $O/r05_d_baselines.txt
timeout 200 python bench.py --batch 100 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch100 %9.1f frames/s  %7.2f ms/step' % (d['value'], d['ms_per_step']))" | tee -a $O/r05_d_baselines.txt
timeout 200 python bench.py --dataset beat --precision fp32 --batch 256 --steps 5 --warmup 2 --no-cpu-baseline --no-chain-latency 2>/dev/null | tail -1 > $O/r05_d_bench_beat_fp32.json; python scripts/bench_brief.py $O/r05_d_bench_beat_fp32.json | tee -a $O/r05_d_baselines.txt
