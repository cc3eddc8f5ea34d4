#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
for ts in 1 0; do
  D=$O/prof_k$ts; rm -rf $D; mkdir -p $D
  DSH_GEMM_TILE=$ts timeout 200 rocprofv3 --kernel-trace -d $D -o p -- python bench.py --dataset beat --precision fp32 --batch 256 --steps 1 --warmup 0 --no-cpu-baseline --no-chain-latency --no-roofline > $D/bench.log 2>&1
  DB=$(find $D -name "*.db" | head -1)
  echo "== DSH_GEMM_TILE=$ts"; python scripts/gemm_shape_summary.py $DB gemm_nt | head -24 | tee $O/k_gemm_shapes_tile$ts.txt
  rm -rf $D
done
