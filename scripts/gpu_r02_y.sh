#!/bin/bash
# extra evidence: rocprofv3 kernel stats of the config-2 (fp32) bench and of the chain-mode bench
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
D=$O/prof_y1; rm -rf $D; mkdir -p $D
timeout 300 rocprofv3 --kernel-trace --stats -d $D -o p -- python bench.py --dataset beat --precision fp32 --batch 256 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-chain-latency > $D/bench.log 2>&1
DB=$(find $D -name "*.db" | head -1)
python scripts/rocprof_summary.py $DB 3 > $O/r02_f_beat_fp32_kernel_stats.txt 2>&1; tail -1 $D/bench.log | cut -c1-300 >> $O/r02_f_beat_fp32_kernel_stats.txt; head -8 $O/r02_f_beat_fp32_kernel_stats.txt; rm -rf $D
D=$O/prof_y2; rm -rf $D; mkdir -p $D
timeout 300 rocprofv3 --kernel-trace --stats -d $D -o p -- python bench.py --mode chain --steps 1 --warmup 1 > $D/bench.log 2>&1
DB=$(find $D -name "*.db" | head -1)
python scripts/rocprof_summary.py $DB 2 > $O/r02_f_chain_mode_kernel_stats.txt 2>&1; tail -1 $D/bench.log | cut -c1-300 >> $O/r02_f_chain_mode_kernel_stats.txt; head -12 $O/r02_f_chain_mode_kernel_stats.txt; rm -rf $D
