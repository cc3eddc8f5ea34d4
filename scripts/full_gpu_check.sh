cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
mkdir -p gpurun_out/prof_r01f
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01f -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_r01f/bench.log 2>&1
find gpurun_out/prof_r01f -name "*.db" | head -3
DB=$(find gpurun_out/prof_r01f -name "*.db" | head -1)
python scripts/rocprof_summary.py $DB 3 > gpurun_out/r01_f_bench_kernel_stats.txt 2>&1
head -30 gpurun_out/r01_f_bench_kernel_stats.txt
tail -2 gpurun_out/prof_r01f/bench.log | cut -c1-400
rm -rf gpurun_out/prof_r01f/*/*.db gpurun_out/prof_r01f/*.db
