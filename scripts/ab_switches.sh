#!/bin/bash
# Same-box A / B lines for the switches DESIGN.md section 4.8 quotes (kept experiments and rejected ones): the default bench
# (SHOW B = 950 ddim25 bf16, 3 timed steps) under each setting -> gpurun_out/<tag>_ab_switches.txt
cd $GRAFT_REPO_ROOT
TAG=${1:-r04}
OUT=gpurun_out/${TAG}_ab_switches.txt
: > $OUT
run() {   # label, env assignments...
  local label="$1"; shift
  local line
  line=$(env "$@" timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 |
         python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f frames/s  %7.2f ms/step' % (d['value'], d['ms_per_step']))")
  printf "%-58s %s\n" "$label" "$line" | tee -a $OUT
}
run "default (3 streams, FFN v3, hi/lo planes, rev order)" X=1
run "default, repeated" X=1
run "DSH_HILO=0 (fp32 residual stream + bf16 shadow)" DSH_HILO=0
run "DSH_FFN_V=2 (round-3 fused FFN kernel; implies fp32 stream)" DSH_FFN_V=2
run "DSH_REV=0 (every launch walks the rows in one order)" DSH_REV=0
run "DSH_STAGGER=4,3 (first-round start stagger)" DSH_STAGGER=4,3
run "DSH_SPLIT_PREFETCH=1 (x-independent head on side streams)" DSH_SPLIT_PREFETCH=1
run "DSH_TL2_PP=1 (deferred-epilogue q|k|v kernel)" DSH_TL2_PP=1
run "DSH_DUAL=2" DSH_DUAL=2
run "DSH_DUAL=4" DSH_DUAL=4
run "DSH_DUAL=0 (one stream)" DSH_DUAL=0
run "DSH_DUAL=0 DSH_STAGGER=4,3" DSH_DUAL=0 DSH_STAGGER=4,3
run "DSH_DUAL=0 DSH_REV=0" DSH_DUAL=0 DSH_REV=0
run "DSH_BENCH_ZERO_DATA=1 (power probe: all-zero operands)" DSH_BENCH_ZERO_DATA=1
run "default, repeated" X=1
