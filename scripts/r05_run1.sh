#!/bin/bash
# round-5 GPU call 1: pipelined phase C of the fused FFN kernel (DSH_FFN_PC=1, default) vs the round-4 loop (DSH_FFN_PC=0):
# op tests, block timelines, default bench A/B on the same box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
rocm-smi --showclocks --showpower 2>/dev/null | head -30 > $O/r05_a_smi_idle.txt
echo "== op tests (PC=1)"; timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "ffn" 2>&1 | tail -5
echo "== timelines"
for pc in 0 1 0 1; do
  DSH_FFN_PC=$pc BENCH_FFN_VERS=3 DSH_HILO=1 timeout 200 python scripts/bench_tl2.py ffn 2>&1 | grep -v amdgpu.ids | sed "s/^/PC=$pc /" | tee -a $O/r05_a_ffn_block_timeline.txt
done
echo "== bench A/B"
for pc in 1 0 1 0; do
  DSH_FFN_PC=$pc timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PC=$pc %9.1f frames/s  %7.2f ms/step  e2e_err %s' % (d['value'], d['ms_per_step'], d.get('bf16_e2e_rel_err')))" | tee -a $O/r05_a_ab.txt
done
echo "== full gpu suite"; timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $O/r05_a_pytest_gpu.txt
