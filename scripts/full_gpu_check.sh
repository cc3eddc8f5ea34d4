#!/bin/bash
# Round-end GPU validation: parity tests, smoke, default bench, rocprofv3 kernel stats of the same command
# (3 untouched steps: --no-roofline skips the instrumented single-stream step and the B=1 latency probe).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r01_f}
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; python scripts/bench_brief.py gpurun_out/${TAG}_bench.json
mkdir -p gpurun_out/prof_$TAG gpurun_out/prof_${TAG}s
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_$TAG/bench.log 2>&1
DB=$(find gpurun_out/prof_$TAG -name "*.db" | head -1)
python scripts/rocprof_summary.py $DB 3 > gpurun_out/${TAG}_bench_kernel_stats.txt 2>&1
tail -1 gpurun_out/prof_$TAG/bench.log | cut -c1-300 >> gpurun_out/${TAG}_bench_kernel_stats.txt
DSH_DUAL=0 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG}s -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_${TAG}s/bench.log 2>&1
DB=$(find gpurun_out/prof_${TAG}s -name "*.db" | head -1)
python scripts/rocprof_summary.py $DB 3 > gpurun_out/${TAG}_single_stream_kernel_stats.txt 2>&1
tail -1 gpurun_out/prof_${TAG}s/bench.log | cut -c1-300 >> gpurun_out/${TAG}_single_stream_kernel_stats.txt
head -12 gpurun_out/${TAG}_bench_kernel_stats.txt; head -12 gpurun_out/${TAG}_single_stream_kernel_stats.txt
rm -rf gpurun_out/prof_$TAG gpurun_out/prof_${TAG}s
