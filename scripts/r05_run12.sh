#!/bin/bash
# round-5 GPU call 12: the hi plane of the residual kept in registers through the fused FFN's last stage (DSH_FFN_PC=3); calibration of the
# FETCH_SIZE / WRITE_SIZE counters against copies of known size
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
echo "== bit identity"; timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "pipelined_phase_c" 2>&1 | tail -5
echo "== block timeline"
for pc in 1 3; do
  echo "-- DSH_FFN_PC=$pc (input = hi plane)" | tee -a $O/r05_n_ffn_kh_timeline.txt
  BENCH_FFN_VERS=3 DSH_HILO=1 DSH_FFN_X_IS_HI=1 DSH_FFN_PC=$pc timeout 200 python scripts/bench_tl2.py ffn 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $O/r05_n_ffn_kh_timeline.txt
done
echo "== bench A/B"
for cfg in "DSH_FFN_PC=3" "DSH_FFN_PC=1" "DSH_FFN_PC=3" "DSH_FFN_PC=1"; do
  env $cfg timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-16s %9.1f frames/s  %7.2f ms/step' % ('$cfg', d['value'], d['ms_per_step']))" | tee -a $O/r05_n_ab_ffn_kh.txt
done
echo "== counter calibration"
P=$O/pmc_cal; rm -rf $P; mkdir -p $P
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/f -o p --output-format csv -- python scripts/pmc_calibrate.py run > $P/f.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/w -o p --output-format csv -- python scripts/pmc_calibrate.py run > $P/w.log 2>&1
python scripts/pmc_calibrate.py parse $P > $O/r05_n_pmc_calibration.txt 2>&1; cat $O/r05_n_pmc_calibration.txt | head -60
rm -rf $P
