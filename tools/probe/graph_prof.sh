# Kernel trace + timeline of the default bench with and without hipGraph replay (MNAS_GRAPHS=1): where does the replayed step lose
# its 0.9 %?   usage (GPU box, repo root): bash tools/probe/graph_prof.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  if [ $v = 1 ]; then export MNAS_GRAPHS=1; else unset MNAS_GRAPHS; fi
  rm -rf /tmp/prof_g$v
  rocprofv3 --kernel-trace --stats -d /tmp/prof_g$v -o kt -- python3 $R/bench.py --steps 10 --warmup 5 --min-seconds 0 --no-cpu-baseline --no-box --no-roofline > $R/gpurun_out/prof_g${v}_bench.log 2>&1
  DB=$(find /tmp/prof_g$v -name "*.db" | head -1)
  python3 $R/tools/rocpd_stats.py $DB 15 > $R/gpurun_out/prof_g${v}_kernel_trace.txt
  python3 $R/tools/rocpd_stats.py --timeline $DB 1500 > $R/gpurun_out/prof_g${v}_timeline.txt
  echo "== graphs=$v"; grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/prof_g${v}_bench.log | head -1
  head -8 $R/gpurun_out/prof_g${v}_timeline.txt; sed -n 2,2p $R/gpurun_out/prof_g${v}_kernel_trace.txt
done
