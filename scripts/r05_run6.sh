#!/bin/bash
# round-5 GPU call 6: the residual-carrying launches (StylizationBlock of the attention branch, feat_proj.3) on the rolling LDS-DMA loop with
# hi / lo planes (DSH_TL2_HL) — tests, same-box A/B, stream count
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
echo "== op tests"; timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "rolling or pipelined or tl_linear or hilo" 2>&1 | tail -3
echo "== instrumented steps (per-class launch times)"
for hl in 1 0; do
  DSH_TL2_HL=$hl timeout 300 python bench.py --no-cpu-baseline --no-chain-latency > $O/r05_f_bench_hl$hl.json 2> $O/r05_f_bench.err; echo "DSH_TL2_HL=$hl"; python scripts/bench_brief.py $O/r05_f_bench_hl$hl.json | grep -v "gemm_nt\|ffn.linear1"
done
echo "== bench A/B"
for cfg in "DSH_TL2_HL=1" "DSH_TL2_HL=0" "DSH_TL2_HL=1" "DSH_TL2_HL=0" "DSH_DUAL=2" "DSH_DUAL=4" "DSH_REV=0"; do
  env $cfg timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-16s %9.1f frames/s  %7.2f ms/step' % ('$cfg', d['value'], d['ms_per_step']))" | tee -a $O/r05_f_ab.txt
done
echo "== full gpu suite"; timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $O/r05_f_pytest_gpu.txt
