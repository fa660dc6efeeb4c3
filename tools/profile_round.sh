#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box: kernel trace of the default bench command, then FETCH_SIZE and
# WRITE_SIZE in two separate --pmc passes (never combined with tracing).  Summaries -> gpurun_out/prof_<tag>_*.txt
# usage (from the repo root on the box): bash tools/profile_round.sh <tag>
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python3 $R/bench.py --steps 10 --warmup 5 --min-seconds 0 --auto-graphs 0 --no-cpu-baseline --no-box > $R/gpurun_out/prof_${TAG}_bench.log 2>&1
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
# 5 warmup + 4 calibration + 10 timed = 19 steps in the trace
python3 $R/tools/rocpd_stats.py $DB 19 > $R/gpurun_out/prof_${TAG}_kernel_trace.txt
python3 $R/tools/rocpd_stats.py --timeline $DB 1500 > $R/gpurun_out/prof_${TAG}_timeline.txt
rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_f -o f -- python3 $R/bench.py --steps 2 --warmup 2 --min-seconds 0 --auto-graphs 0 --no-cpu-baseline --no-box --no-roofline > $R/gpurun_out/prof_${TAG}_pmc_fetch.log 2>&1
DB=$(find /tmp/prof_f -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py --pmc $DB 4 > $R/gpurun_out/prof_${TAG}_pmc_fetch_size.txt
rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_w -o w -- python3 $R/bench.py --steps 2 --warmup 2 --min-seconds 0 --auto-graphs 0 --no-cpu-baseline --no-box --no-roofline > $R/gpurun_out/prof_${TAG}_pmc_write.log 2>&1
DB=$(find /tmp/prof_w -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py --pmc $DB 4 > $R/gpurun_out/prof_${TAG}_pmc_write_size.txt
