#!/bin/bash
# Round-6 (second session) GPU calls: scripts/r06b_run.sh <tag> <step> [<step> ...]; outputs under gpurun_out/<tag>_*
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out
F32="--dataset beat --precision fp32 --batch 256 --no-cpu-baseline --no-chain-latency"
for step in "$@"; do
  case $step in
    f32test)  timeout 1200 python -m pytest tests/test_gpu_gemm_f32_pro.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -30 > $O/${TAG}_f32test.txt; cat $O/${TAG}_f32test.txt ;;
    f32gold)  timeout 1800 python -m pytest tests/test_gpu_eval.py tests/test_gpu_sampler.py tests/test_gpu_ops_golden.py -x -q -k "fp32 or beat or config2" 2>&1 | grep -v amdgpu.ids | tail -8 > $O/${TAG}_f32gold.txt; cat $O/${TAG}_f32gold.txt ;;
    f32ab:*)  # f32ab:<ENVVAR>:<v0>,<v1>  alternating fp32 config-2 bench runs
              spec=${step#f32ab:}; var=${spec%%:*}; vals=${spec#*:}
              for rep in 1 2; do for v in ${vals//,/ }; do
                env $var=$v timeout 300 python bench.py $F32 --steps 5 --warmup 2 --no-roofline 2>/dev/null | tail -1 > $O/.ab.json
                python - <<PY >> $O/${TAG}_f32ab_${var}.txt
import json; d = json.load(open("$O/.ab.json")); print("$var=$v", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step", d.get("telemetry", {}).get("clock_mhz_mean"), "MHz", d.get("telemetry", {}).get("power_w_mean"), "W")
PY
              done; done; cat $O/${TAG}_f32ab_${var}.txt ;;
    f32stats:*) # f32stats:<name>:<ENV=val>  rocprofv3 kernel stats of the fp32 config-2 bench on one stream
              spec=${step#f32stats:}; nm=${spec%%:*}; ev=${spec#*:}
              D=$O/prof_${TAG}_$nm; rm -rf $D; mkdir -p $D
              env DSH_DUAL=0 $ev timeout 400 rocprofv3 --kernel-trace --stats -d $D -o p -- python bench.py $F32 --steps 2 --warmup 1 --no-roofline > $D/bench.log 2>&1
              DB=$(find $D -name "*.db" | head -1)
              python scripts/rocprof_summary.py $DB 3 > $O/${TAG}_${nm}_kernel_stats.txt 2>&1
              tail -1 $D/bench.log | cut -c1-200 >> $O/${TAG}_${nm}_kernel_stats.txt
              head -16 $O/${TAG}_${nm}_kernel_stats.txt; rm -rf $D ;;
    f32micro) timeout 300 python scripts/bench_f32_gemm.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_f32micro.txt; cat $O/${TAG}_f32micro.txt ;;
    f32abl)   for a in 0 1 2 4 8 16 3 7 ; do echo "== DSH_GP_ABL=$a"; DSH_GP_ABL=$a timeout 120 python scripts/bench_f32_gemm.py 2>&1 | grep "pro0"; done > $O/${TAG}_f32abl.txt; cat $O/${TAG}_f32abl.txt ;;
    cfg1ab)   for rep in 1 2; do for v in 0 7; do echo "DSH_F32_FUSE=$v $(DSH_F32_FUSE=$v timeout 300 python scripts/run_config1.py 2>&1 | tail -1)"; done; done > $O/${TAG}_cfg1ab.txt; cat $O/${TAG}_cfg1ab.txt ;;
    f32cold)  timeout 300 python scripts/bench_f32_cold.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_f32cold.txt; cat $O/${TAG}_f32cold.txt ;;
    f32dma)   for v in 0 1; do echo "== DSH_GP_DMA=$v"; DSH_GP_DMA=$v timeout 200 python scripts/bench_f32_gemm.py 2>&1 | grep "pro"; DSH_GP_DMA=$v timeout 200 python scripts/bench_f32_cold.py 2>&1 | grep "pro0"; done > $O/${TAG}_f32dma.txt; cat $O/${TAG}_f32dma.txt ;;
    f32ab2)   # DMA x fuse bits in the fp32 config-2 step
              for rep in 1 2; do for cfg in "0 7" "1 7" "1 5" "1 4" "1 6"; do set -- $cfg
                DSH_GP_DMA=$1 DSH_F32_FUSE=$2 timeout 300 python bench.py $F32 --steps 5 --warmup 2 --no-roofline 2>/dev/null | tail -1 > $O/.ab.json
                python - <<PY >> $O/${TAG}_f32ab2.txt
import json; d = json.load(open("$O/.ab.json")); print("DSH_GP_DMA=$1 DSH_F32_FUSE=$2", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step", round(d.get("telemetry", {}).get("clock_mhz_mean", 0)), "MHz", round(d.get("telemetry", {}).get("power_w_mean", 0)), "W")
PY
              done; done; cat $O/${TAG}_f32ab2.txt ;;
    midab)    # mid-size regimes vs the sub-batch stream policy
              for rep in 1 2; do for cfg in "DSH_DUAL=0" "DSH_DUAL=3" "DSH_DUAL_ROWS=6000" "DSH_DUAL_ROWS=4500"; do
                env $cfg timeout 200 python bench.py --batch 100 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 > $O/.ab.json
                python - <<PY >> $O/${TAG}_midab.txt
import json; d = json.load(open("$O/.ab.json")); print("b100 $cfg", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step")
PY
              done; done
              for cfg in "DSH_DUAL=3" "DSH_DUAL_ROWS=14000" "DSH_DUAL_ROWS=11000"; do
                env $cfg timeout 400 python bench.py --mode ddpm --batch 313 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 > $O/.ab.json
                python - <<PY >> $O/${TAG}_midab.txt
import json; d = json.load(open("$O/.ab.json")); print("ddpm313 $cfg", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step")
PY
              done; cat $O/${TAG}_midab.txt ;;
    midffn)   for b in 60 100 160 240; do for v in 1 0; do
                DSH_FFN_FUSE=$v timeout 200 python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 > $O/.ab.json
                python - <<PY >> $O/${TAG}_midffn.txt
import json; d = json.load(open("$O/.ab.json")); print("batch $b DSH_FFN_FUSE=$v", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step")
PY
              done; done; cat $O/${TAG}_midffn.txt ;;
    midsplit) for b in 100 160; do for cfg in "DSH_DUAL_MIN_ROWS=12288" "DSH_DUAL_MIN_ROWS=4000" "DSH_DUAL_MIN_ROWS=4000 DSH_DUAL_ROWS=3000"; do
                env $cfg timeout 200 python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 > $O/.ab.json
                python - <<PY >> $O/${TAG}_midsplit.txt
import json; d = json.load(open("$O/.ab.json")); print("batch $b $cfg", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step")
PY
              done; done; cat $O/${TAG}_midsplit.txt ;;
    f32pmc)   # PMC counters of one single-stream fp32 config-2 step (separate passes, as MI355X_MICROARCH.md prescribes)
              export DSH_DUAL=0; P=$O/pmc_${TAG}; rm -rf $P; mkdir -p $P
              for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
                tag=$(echo $set | tr ' ' '_' | cut -c1-40)
                timeout 300 rocprofv3 --kernel-trace --pmc $set -d $P/$tag -o p --output-format csv -- python bench.py $F32 --steps 1 --warmup 0 --no-roofline > $P/$tag.log 2>&1
              done
              python scripts/pmc_to_json.py $P $O/${TAG}_pmc_step_beat_fp32.json gemm_f32_pro_kernel gemm_nt_kernel linear_attention_pre_kernel gemm_nt_ksplit_kernel
              python - <<PY
import json; f = "$O/${TAG}_pmc_step_beat_fp32.json"; d = json.load(open(f))
d["source"] = d["source"].replace("the default bench (SHOW B=950 T=88 CFG ddim25 bf16)", "the fp32 parity configuration (BEAT B=256 T=34 ddim25 fp32, BASELINE configs[1])").replace("scripts/gpu_profiles.sh", "scripts/r06b_run.sh f32pmc")
json.dump(d, open(f, "w"), indent=1)
for k, e in d["kernels"].items(): print(k[:60], {q: (round(v, 3) if isinstance(v, float) else v) for q, v in e.items() if q in ("hbm_traffic_bytes", "l2_hit_rate", "mfma_busy_frac")}, e.get("wave_cycles_breakdown"))
PY
              rm -rf $P; unset DSH_DUAL ;;
    f32attn)  for rep in 1 2; do for v in 0 1; do
                DSH_ATTN_F32_MFMA=$v timeout 300 python bench.py $F32 --steps 5 --warmup 2 --no-roofline 2>/dev/null | tail -1 > $O/.ab.json
                python - <<PY >> $O/${TAG}_f32attn.txt
import json; d = json.load(open("$O/.ab.json")); print("DSH_ATTN_F32_MFMA=$v", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step")
PY
              done; done; cat $O/${TAG}_f32attn.txt ;;
    f32bits)  for rep in 1 2; do for v in 7 15; do
                DSH_F32_FUSE=$v timeout 300 python bench.py $F32 --steps 5 --warmup 2 --no-roofline 2>/dev/null | tail -1 > $O/.ab.json
                python - <<PY >> $O/${TAG}_f32bits.txt
import json; d = json.load(open("$O/.ab.json")); print("DSH_F32_FUSE=$v", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step")
PY
              done; done; cat $O/${TAG}_f32bits.txt ;;
    pipeab)   for rep in 1 2; do for v in 0 1; do
                DSH_PIPE=$v timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/.ab.json
                python - <<PY >> $O/${TAG}_pipeab.txt
import json; d = json.load(open("$O/.ab.json")); c = d.get("chain_window_latency", {})
print("DSH_PIPE=$v", {k: {q: round(v, 2) for q, v in c[k].items() if q.startswith("p50") or q.startswith("frames")} for k in c if k.startswith("chains")})
PY
                DSH_PIPE=$v timeout 300 python bench.py --mode chain --steps 2 --warmup 1 2>/dev/null | tail -1 > $O/.ab.json
                python - <<PY >> $O/${TAG}_pipeab.txt
import json; d = json.load(open("$O/.ab.json")); print("DSH_PIPE=$v chain-mode", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step")
PY
              done; done; cat $O/${TAG}_pipeab.txt ;;
    piperows) for b in 24 46 60 100 130; do for cfg in "DSH_PIPE_ROWS=4096" "DSH_PIPE_ROWS=12287"; do
                env $cfg timeout 200 python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 > $O/.ab.json
                python - <<PY >> $O/${TAG}_piperows.txt
import json; d = json.load(open("$O/.ab.json")); print("batch $b $cfg", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step")
PY
              done; done; cat $O/${TAG}_piperows.txt ;;
    piperows2) for b in 160 200 240 313; do for cfg in "DSH_PIPE_ROWS=12287" "DSH_PIPE_ROWS=40000 DSH_DUAL_MIN_ROWS=40000"; do
                env $cfg timeout 200 python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 > $O/.ab.json
                python - <<PY >> $O/${TAG}_piperows2.txt
import json; d = json.load(open("$O/.ab.json")); print("batch $b $cfg", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step")
PY
              done; done; cat $O/${TAG}_piperows2.txt ;;
    piperows3) for b in 475 650 950; do for cfg in "DSH_PIPE_ROWS=12287" "DSH_PIPE_ROWS=100000 DSH_DUAL_MIN_ROWS=100000"; do
                env $cfg timeout 300 python bench.py --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 > $O/.ab.json
                python - <<PY >> $O/${TAG}_piperows3.txt
import json; d = json.load(open("$O/.ab.json")); print("batch $b $cfg", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step", d.get("telemetry", {}).get("clock_mhz_mean"), d.get("telemetry", {}).get("power_w_mean"))
PY
              done; done; cat $O/${TAG}_piperows3.txt ;;
    f32pipe)  for rep in 1 2; do for cfg in "DSH_PIPE_ROWS=12287" "DSH_PIPE_ROWS=100000 DSH_DUAL_MIN_ROWS=100000"; do
                env $cfg timeout 300 python bench.py $F32 --steps 5 --warmup 2 --no-roofline 2>/dev/null | tail -1 > $O/.ab.json
                python - <<PY >> $O/${TAG}_f32pipe.txt
import json; d = json.load(open("$O/.ab.json")); print("fp32 B=256 $cfg", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step")
PY
              done; done; cat $O/${TAG}_f32pipe.txt ;;
    ddpmpipe) for cfg in "DSH_PIPE=0" "DSH_PIPE=1"; do
                env $cfg timeout 400 python bench.py --mode ddpm --batch 313 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 > $O/.ab.json
                python - <<PY >> $O/${TAG}_ddpmpipe.txt
import json; d = json.load(open("$O/.ab.json")); print("ddpm313 $cfg", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step")
PY
                echo "config1 $cfg $(env $cfg timeout 300 python scripts/run_config1.py 2>&1 | tail -1)" >> $O/${TAG}_ddpmpipe.txt
              done; cat $O/${TAG}_ddpmpipe.txt ;;
    graphrows) for rep in 1 2; do for v in 4096 8192; do
                DSH_GRAPH_ROWS=$v timeout 300 python bench.py --mode chain --steps 2 --warmup 1 2>/dev/null | tail -1 > $O/.ab.json
                python - <<PY >> $O/${TAG}_graphrows.txt
import json; d = json.load(open("$O/.ab.json")); print("DSH_GRAPH_ROWS=$v chain-mode", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step")
PY
              done; done; cat $O/${TAG}_graphrows.txt ;;
    tlsrows)  for v in 0 2048 4096 6144 9000; do
                DSH_TLS_ROWS=$v timeout 300 python bench.py --mode chain --steps 2 --warmup 1 2>/dev/null | tail -1 > $O/.ab.json
                python - <<PY >> $O/${TAG}_tlsrows.txt
import json; d = json.load(open("$O/.ab.json")); print("DSH_TLS_ROWS=$v chain-mode", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step")
PY
              done; cat $O/${TAG}_tlsrows.txt ;;
    headpipe) for rep in 1 2 3; do for cfg in "DSH_PIPE_ROWS=64499" "DSH_PIPE_ROWS=100000"; do
                env $cfg timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 > $O/.ab.json
                python - <<PY >> $O/${TAG}_headpipe.txt
import json; d = json.load(open("$O/.ab.json")); print("batch 950 $cfg", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step", round(d.get("telemetry", {}).get("clock_mhz_mean", 0)), "MHz", round(d.get("telemetry", {}).get("power_w_mean", 0)), "W")
PY
              done; done; cat $O/${TAG}_headpipe.txt ;;
    prioab)   for rep in 1 2; do for v in 0 1; do   # (historical: the switch was removed after this measurement)
                DSH_PIPE_PRIO=$v timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/.ab.json
                python - <<PY >> $O/${TAG}_prioab.txt
import json; d = json.load(open("$O/.ab.json")); c = d.get("chain_window_latency", {})
print("DSH_PIPE_PRIO=$v", {k: {q: round(v, 2) for q, v in c[k].items() if q.startswith("p50")} for k in c if k.startswith("chains")})
PY
                DSH_PIPE_PRIO=$v timeout 300 python bench.py --mode chain --steps 2 --warmup 1 2>/dev/null | tail -1 > $O/.ab.json
                python - <<PY >> $O/${TAG}_prioab.txt
import json; d = json.load(open("$O/.ab.json")); print("DSH_PIPE_PRIO=$v chain-mode", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step")
PY
                DSH_PIPE_PRIO=$v timeout 200 python bench.py --batch 100 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 > $O/.ab.json
                python - <<PY >> $O/${TAG}_prioab.txt
import json; d = json.load(open("$O/.ab.json")); print("DSH_PIPE_PRIO=$v batch 100", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step")
PY
              done; done; cat $O/${TAG}_prioab.txt ;;
    f32bench) timeout 300 python bench.py $F32 2>/dev/null | tail -1 > $O/${TAG}_bench_beat_fp32.json; python scripts/bench_brief.py $O/${TAG}_bench_beat_fp32.json ;;
    *)        bash scripts/r06_run.sh $TAG $step ;;
  esac
done
