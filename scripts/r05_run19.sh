#!/bin/bash
# round-5 GPU call 19: tile shape of the small bf16 GEMMs (encoder_aud, audio_proj, embeddings): DSH_GEMM_TILE = 2 (128 x 64) / 0 (128 x 128) / 4 (64 x 32) vs the default 64 x 64
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
for cfg in "DSH_GEMM_TILE=1" "DSH_GEMM_TILE=2" "DSH_GEMM_TILE=4" "DSH_GEMM_TILE=0" "DSH_GEMM_TILE=1" "DSH_GEMM_TILE=2"; do
  env $cfg timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-chain-latency 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; g=[v for n,v in k.items() if n.startswith('gemm_nt')][0]
print('%-18s %9.1f frames/s  %7.2f ms/step   small GEMMs (instrumented single-stream step): %6.2f ms/step, %5.1f us per launch' % ('$cfg', d['value'], d['ms_per_step'], g['ms_per_step'], g['avg_launch_us']))" | tee -a $O/r05_u_ab_gemm_tile.txt
done
