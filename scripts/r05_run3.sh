#!/bin/bash
# round-5 GPU call 3: the new tests, first-round start stagger on the pipelined FFN kernel in isolation (is pass B at the HBM wall
# because every CU is in it at once?), telemetry in the bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_ops_golden.py tests/test_gpu_ops.py tests/test_gpu_sharded.py -q -m gpu -s -k "fixtures or rolling or pipelined or world_size_8" 2>&1 | grep -v amdgpu.ids | grep "ops_show\|world 8\|passed\|failed\|Error\|error" | tee $O/r05_c_new_tests.txt
echo "== FFN stagger in isolation"
for sg in "0,0" "4,1" "4,2" "8,1" "8,2" "2,4"; do
  DSH_STAGGER=$sg BENCH_FFN_VERS=3 DSH_HILO=1 timeout 200 python scripts/bench_tl2.py ffn 2>&1 | grep -v amdgpu.ids | sed "s/^/STAGGER=$sg /" | tee -a $O/r05_c_ffn_stagger_timeline.txt
done
echo "== default bench (telemetry)"
timeout 400 python bench.py --no-cpu-baseline --no-chain-latency > $O/r05_c_bench.json 2> $O/r05_c_bench.err; python scripts/bench_brief.py $O/r05_c_bench.json
python - <<'PY'
import json; d=json.load(open("gpurun_out/r05_c_bench.json")); print({k:d.get(k) for k in ("value","ms_per_step","host_enqueue_ms_per_step","host_affinity","telemetry","end_to_end_mfma_frac","end_to_end_mfma_frac_at_measured_clock")})
PY
echo "== full gpu suite"; timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $O/r05_c_pytest_gpu.txt
