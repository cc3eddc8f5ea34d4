#!/bin/bash
# round-5 GPU call 11: the attention branch's StylizationBlock as the first stage of the fused FFN launch (DSH_FFN_STY); default bench in full (timing)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
echo "== bit identity"; timeout 600 python -m pytest tests/test_gpu_eval.py -x -q -m gpu -k "fused_attention_branch" 2>&1 | tail -8
echo "== bench A/B"
for cfg in "DSH_FFN_STY=1" "DSH_FFN_STY=0" "DSH_FFN_STY=1" "DSH_FFN_STY=0"; do
  env $cfg timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-16s %9.1f frames/s  %7.2f ms/step' % ('$cfg', d['value'], d['ms_per_step']))" | tee -a $O/r05_k_ab_sty.txt
done
echo "== default bench exactly as the driver runs it"
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_k_bench_driver.json 2> $O/r05_k_bench_driver.err ) 2>&1 | tail -3
python scripts/bench_brief.py $O/r05_k_bench_driver.json | head -12
