# Kernel trace of the squeeze-excite variant (bench.py --se) -> gpurun_out/prof_se_kernel_trace.txt; prints the top of the table.
# usage (GPU box, from the repo root): bash tools/probe/se_prof.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_se -o kt -- python3 $R/bench.py --se --steps 10 --warmup 5 --min-seconds 0 --no-cpu-baseline --no-box > $R/gpurun_out/prof_se_bench.log 2>&1
DB=$(find /tmp/prof_se -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB 19 > $R/gpurun_out/prof_se_kernel_trace.txt
head -45 $R/gpurun_out/prof_se_kernel_trace.txt | cut -c1-60,88-160
