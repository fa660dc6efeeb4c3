#!/bin/bash
# quick kernel-trace summary of a short bench run on the GPU box: bash tools/prof_quick.sh [grep pattern]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 280 rocprofv3 --kernel-trace --stats -d /tmp/prof_q -o q -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_quick.log 2>&1
DB=$(find /tmp/prof_q -name "*.db" | head -1)
if [ -z "$DB" ]; then echo "no trace db"; tail -5 $R/gpurun_out/prof_quick.log; exit 1; fi
# 3 warmup + 4 calibration + 5 timed = 12 steps
python3 $R/tools/rocpd_stats.py $DB 12 > $R/gpurun_out/prof_quick_kernel_trace.txt; python3 $R/tools/rocpd_stats.py --seq $DB > $R/gpurun_out/prof_quick_seq.txt
if [ -n "$1" ]; then grep -E "$1" $R/gpurun_out/prof_quick_kernel_trace.txt | cut -c1-170; else head -40 $R/gpurun_out/prof_quick_kernel_trace.txt | cut -c1-170; fi
