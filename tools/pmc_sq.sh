#!/bin/bash
# SQ-level counters per kernel for the bench step (two --pmc passes, never combined with tracing).  usage: bash tools/pmc_sq.sh <tag>
TAG=${1:-sq}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/pmc_${TAG}_available.txt
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
P2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SALU"
n=1
for P in "$P1" "$P2"; do
  rocprofv3 --pmc $P -d /tmp/prof_sq$n -o sq -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_${TAG}_$n.log 2>&1
  DB=$(find /tmp/prof_sq$n -name "*.db" | head -1)
  python3 $R/tools/rocpd_stats.py --pmc $DB 4 > $R/gpurun_out/pmc_${TAG}_$n.txt
  n=$((n+1))
done
