#!/bin/bash
# PMC passes over ONE single-stream bench step (all kernels at their real shapes) -> gpurun_out/r01_pmc_step.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_step; mkdir -p $OUT
export DSH_DUAL=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set -d $OUT/$tag -o p --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $OUT/$tag.log 2>&1
done
python scripts/pmc_to_json.py $OUT gpurun_out/r01_pmc_step.json tl_linear_kernel linear_attention_tiled gemm_nt_kernel seed_stream
rm -rf $OUT/*/
