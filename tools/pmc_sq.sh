#!/bin/bash
# SQ-level counters per kernel for the bench step (two --pmc passes, never combined with tracing).  usage: bash tools/pmc_sq.sh <tag>
TAG=${1:-sq}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/pmc_${TAG}_available.txt
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
P2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SALU"
# pass 3: matrix-core evidence (north_star: "MFMA-busy counters"): SQ_VALU_MFMA_BUSY_CYCLES counts cycles the matrix pipe of a SIMD
# is busy (16 per v_mfma_f32_16x16x32_bf16), SQ_BUSY_CU_CYCLES the cycles a CU has work: their ratio / 4 SIMDs = MFMA-busy fraction
P3="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES"
n=1
for P in "$P1" "$P2" "$P3"; do
  rocprofv3 --pmc $P -d /tmp/prof_sq$n -o sq -- python3 $R/bench.py --steps 2 --warmup 2 --min-seconds 0 --auto-graphs 0 --no-cpu-baseline --no-box --no-roofline > $R/gpurun_out/pmc_${TAG}_$n.log 2>&1
  DB=$(find /tmp/prof_sq$n -name "*.db" | head -1)
  python3 $R/tools/rocpd_stats.py --pmc $DB 4 > $R/gpurun_out/pmc_${TAG}_$n.txt
  n=$((n+1))
done
