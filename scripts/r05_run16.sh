#!/bin/bash
# round-5 GPU call 16: smoke() and the bench exactly as the driver runs it, on the final build
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | tail -6
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_r_bench_driver.json 2> $O/r05_r_bench_driver.err ) 2>&1 | tail -3
python scripts/bench_brief.py $O/r05_r_bench_driver.json | head -3
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_r_bench_driver.json"))
print({k:d[k] for k in ("value","ms_per_step","end_to_end_mfma_frac","clock_mhz_mean","power_w_mean","kernel_build_id")})
r=d["roofline"]; print({k:r[k] for k in ("achieved","frac","traffic","avg_launch_us")})
PY
