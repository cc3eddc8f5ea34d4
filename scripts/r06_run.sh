#!/bin/bash
# Round-6 GPU calls, one parameterised script: scripts/r06_run.sh <tag> <step> [<step> ...]   (steps run in order; outputs under gpurun_out/<tag>_*)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out
for step in "$@"; do
  case $step in
    tl4test)  timeout 900 python -m pytest tests/test_gpu_tl4.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -25 > $O/${TAG}_tl4test.txt; tail -12 $O/${TAG}_tl4test.txt ;;
    advtests) timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "rolling" 2>&1 | tail -5 > $O/${TAG}_advtests.txt; cat $O/${TAG}_advtests.txt ;;
    tl4bench) timeout 600 python scripts/bench_tl4.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_tl4bench.txt; cat $O/${TAG}_tl4bench.txt
              timeout 300 python scripts/bench_tl4.py 27896 2>&1 | grep -v amdgpu.ids > $O/${TAG}_tl4bench_third.txt; cat $O/${TAG}_tl4bench_third.txt ;;
    ab:*)     # ab:<ENVVAR>:<v0>,<v1>,...  alternating default-bench runs (3 timed steps) under each value of one switch
              spec=${step#ab:}; var=${spec%%:*}; vals=${spec#*:}
              for rep in 1 2; do for v in ${vals//,/ }; do
                env $var=$v timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-chain-latency --no-roofline 2>/dev/null | tail -1 > $O/.ab.json
                python - <<PY >> $O/${TAG}_ab_${var}.txt
import json; d = json.load(open("$O/.ab.json")); print("$var=$v", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step", d.get("telemetry", {}).get("clock_mhz_mean"), "MHz", d.get("telemetry", {}).get("power_w_mean"), "W")
PY
              done; done; cat $O/${TAG}_ab_${var}.txt ;;
    regimes)  # the other regimes of DESIGN 4.5: chain latencies, config 4's per-GPU share, mid-size batch, fp32 config 2
              timeout 300 python bench.py --mode chain --steps 2 --warmup 1 2>/dev/null | tail -1 > $O/${TAG}_bench_chain.json
              timeout 400 python bench.py --mode ddpm --batch 313 --steps 1 --warmup 0 2>/dev/null | tail -1 > $O/${TAG}_bench_ddpm313.json
              timeout 200 python bench.py --batch 100 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/${TAG}_bench_b100.json
              timeout 200 python bench.py --dataset beat --precision fp32 --batch 256 --no-cpu-baseline --no-chain-latency 2>/dev/null | tail -1 > $O/${TAG}_bench_beat_fp32.json
              python - <<PY
import json
for f in ("bench_chain", "bench_ddpm313", "bench_b100", "bench_beat_fp32"):
    try:
        d = json.load(open("$O/${TAG}_" + f + ".json")); print(f, round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step", {k: d[k] for k in ("chain_latency_ms",) if k in d})
    except Exception as e: print(f, "ERR", e)
PY
              ;;
    stats1)   # rocprofv3 kernel stats of the default bench on ONE stream (3 steps)
              D=$O/prof_${TAG}_single; rm -rf $D; mkdir -p $D
              DSH_DUAL=0 timeout 300 rocprofv3 --kernel-trace --stats -d $D -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-chain-latency > $D/bench.log 2>&1
              DB=$(find $D -name "*.db" | head -1)
              python scripts/rocprof_summary.py $DB 3 > $O/${TAG}_single_stream_kernel_stats.txt 2>&1
              tail -1 $D/bench.log | cut -c1-300 >> $O/${TAG}_single_stream_kernel_stats.txt
              head -30 $O/${TAG}_single_stream_kernel_stats.txt; rm -rf $D ;;
    abmode:*) # abmode:<ENVVAR>:<v0>,<v1>:<bench args with _ for spaces>
              spec=${step#abmode:}; var=${spec%%:*}; rest=${spec#*:}; vals=${rest%%:*}; args=${rest#*:}; args=${args//_/ }
              for rep in 1 2; do for v in ${vals//,/ }; do
                env $var=$v timeout 400 python bench.py $args --no-cpu-baseline --no-chain-latency --no-roofline 2>/dev/null | tail -1 > $O/.ab.json
                python - <<PY >> $O/${TAG}_abmode_${var}.txt
import json; d = json.load(open("$O/.ab.json")); print("$args $var=$v", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step")
PY
              done; done; cat $O/${TAG}_abmode_${var}.txt ;;
    statsb:*) # rocprofv3 kernel stats of a bench command: statsb:<name>:<bench args with _ for spaces>
              spec=${step#statsb:}; nm=${spec%%:*}; args=${spec#*:}; args=${args//_/ }
              D=$O/prof_${TAG}_$nm; rm -rf $D; mkdir -p $D
              timeout 400 rocprofv3 --kernel-trace --stats -d $D -o p -- python bench.py $args --no-cpu-baseline --no-roofline --no-chain-latency > $D/bench.log 2>&1
              DB=$(find $D -name "*.db" | head -1)
              python scripts/rocprof_summary.py $DB 1 > $O/${TAG}_${nm}_kernel_stats.txt 2>&1
              tail -1 $D/bench.log | cut -c1-200 >> $O/${TAG}_${nm}_kernel_stats.txt
              head -24 $O/${TAG}_${nm}_kernel_stats.txt; rm -rf $D ;;
    bench)    timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"; python scripts/bench_brief.py $O/${TAG}_bench.json ;;
    suite)    timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -40 > $O/${TAG}_pytest_gpu.txt; tail -5 $O/${TAG}_pytest_gpu.txt ;;
    profiles) bash scripts/gpu_profiles.sh $TAG ;;
    *)        echo "unknown step $step" ;;
  esac
done
