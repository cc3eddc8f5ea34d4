#!/bin/bash
# Evidence run (per round: TAG = r03_e ...): bench JSONs (default with CPU baselines; chain / ddpm modes; config 2), rocprofv3 kernel stats of the
# bench command (default sub-batch streams / one stream), PMC traffic of one single-stream step, kernel trace of a batch-1 chained window.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r02}
O=gpurun_out
timeout 600 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"; python scripts/bench_brief.py $O/${TAG}_bench.json
timeout 200 python bench.py --mode chain --steps 2 --warmup 1 > $O/${TAG}_bench_chain.json 2>> $O/${TAG}_bench.err; echo "chain rc=$?"
timeout 300 python bench.py --mode ddpm --batch 313 --steps 1 --warmup 0 > $O/${TAG}_bench_ddpm313.json 2>> $O/${TAG}_bench.err; echo "ddpm rc=$?"
timeout 200 python bench.py --dataset beat --precision fp32 --batch 256 --no-cpu-baseline --no-chain-latency > $O/${TAG}_bench_beat_fp32.json 2>> $O/${TAG}_bench.err; echo "cfg2 rc=$?"
python - <<PY
import json
for f in ("${TAG}_bench", "${TAG}_bench_chain", "${TAG}_bench_ddpm313", "${TAG}_bench_beat_fp32"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json")); print(f, round(d["value"], 1), d["unit"], round(d["ms_per_step"], 1), "ms/step", d.get("end_to_end_mfma_frac"))
    except Exception as e: print(f, "ERR", e)
PY
for mode in multi single; do
  D=$O/prof_${TAG}_$mode; rm -rf $D; mkdir -p $D
  if [ $mode = single ]; then export DSH_DUAL=0; else unset DSH_DUAL; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $D -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-chain-latency > $D/bench.log 2>&1
  DB=$(find $D -name "*.db" | head -1)
  python scripts/rocprof_summary.py $DB 3 > $O/${TAG}_${mode}_stream_kernel_stats.txt 2>&1
  tail -1 $D/bench.log | cut -c1-300 >> $O/${TAG}_${mode}_stream_kernel_stats.txt
  head -14 $O/${TAG}_${mode}_stream_kernel_stats.txt
  rm -rf $D
done
export DSH_DUAL=0
P=$O/pmc_${TAG}; rm -rf $P; mkdir -p $P
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $P/$tag -o p --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-chain-latency > $P/$tag.log 2>&1
done
python scripts/pmc_to_json.py $P $O/${TAG}_pmc_step.json tl_linear_kernel tl2_linear_kernel tl2_ffn_kernel tl3_ffn_kernel tls_linear_kernel linear_attention_tiled gemm_nt_kernel gemv_rows tl_aud_tail_kernel tl_joint_kernel tl_aproj_kernel cfg_mix_kernel
rm -rf $P
unset DSH_DUAL
bash scripts/prof_chain.sh $TAG 1 2>&1 | grep -v simple_timer | head -14
timeout 200 python scripts/bench_tl2.py ffn 2>&1 | grep -v amdgpu.ids > $O/${TAG}_ffn_block_timeline.txt
BENCH_FFN_VERS=3 DSH_HILO=1 DSH_FFN_X_IS_HI=1 timeout 200 python scripts/bench_tl2.py ffn 2>&1 | grep -v amdgpu.ids >> $O/${TAG}_ffn_block_timeline.txt
tail -2 $O/${TAG}_ffn_block_timeline.txt
