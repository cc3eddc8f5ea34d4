#!/bin/bash
# PMC passes (FETCH_SIZE / WRITE_SIZE / L2 hit / SQ) for the dominant token-per-lane kernels on the bench shapes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_tl2; mkdir -p $OUT
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set -d $OUT/$tag -o p --output-format csv -- python scripts/bench_tl.py 1,2 > $OUT/$tag.log 2>&1
done
python scripts/pmc_summary.py $OUT
python scripts/pmc_to_json.py $OUT gpurun_out/r01_pmc_tl_tiled.json
