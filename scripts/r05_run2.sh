#!/bin/bash
# round-5 GPU call 2: rolling main loop of the q|k|v / feat_proj.1 Linears (DSH_TL2_ROLL) + the reference op fixtures on the GPU
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>/dev/null | head -40 > $O/r05_b_hwmon_ls.txt
for f in /sys/class/drm/card*/device/hwmon/hwmon*/{freq1_input,power1_average,power1_input,freq1_label}; do echo "$f: $(cat $f 2>/dev/null)"; done >> $O/r05_b_hwmon_ls.txt
echo "== new tests"; timeout 600 python -m pytest tests/test_gpu_ops_golden.py tests/test_gpu_ops.py -x -q -m gpu -s -k "golden or fixtures or rolling or pipelined" 2>&1 | grep -v amdgpu.ids | tail -25 | tee $O/r05_b_new_tests.txt
echo "== kernels in isolation"
for r in 0 1 0 1; do
  DSH_TL2_ROLL=$r timeout 300 python scripts/bench_tl2.py qkv,feat1p0,ffn2 2>&1 | grep -v amdgpu.ids | grep -v "ablation\|probe" | sed "s/^/ROLL=$r /" | tee -a $O/r05_b_tl2_roll_timings.txt
done
echo "== bench A/B"
for r in 1 0 1 0; do
  DSH_TL2_ROLL=$r timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ROLL=$r %9.1f frames/s  %7.2f ms/step' % (d['value'], d['ms_per_step']))" | tee -a $O/r05_b_ab.txt
done
echo "== full gpu suite"; timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $O/r05_b_pytest_gpu.txt
