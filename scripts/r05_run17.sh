#!/bin/bash
# round-5 GPU call 17: the power probe of the final build — the default bench with all-zero weights and inputs (same launches, same
# instruction stream, no data-dependent switching) against the real run, alternating, with the sampled clock / power of each
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
for cfg in "DSH_BENCH_ZERO_DATA=0" "DSH_BENCH_ZERO_DATA=1" "DSH_BENCH_ZERO_DATA=0" "DSH_BENCH_ZERO_DATA=1"; do
  env $cfg timeout 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --no-chain-latency 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); t=d.get('telemetry') or {}
print('%-24s %9.1f frames/s  %7.2f ms/step  sclk %6.0f MHz (min %4.0f max %4.0f)  %6.0f W  build %s' % ('$cfg', d['value'], d['ms_per_step'], t.get('clock_mhz_mean',0), t.get('clock_mhz_min',0), t.get('clock_mhz_max',0), t.get('power_w_mean',0), d.get('kernel_build_id')))" | tee -a $O/r05_s_zero_data_probe.txt
done
