#!/bin/bash
# Everything the round's profiles/ directory is built from, in ONE gpurun call:  bash tools/collect_round.sh <tag>
# (kernel trace + timeline, FETCH / WRITE PMC passes, SQ + MFMA counter passes, the default bench line with its CPU baseline,
# the variant configurations).  Summaries land in gpurun_out/; copy the ones to be judged into profiles/.
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
python3 $R/bench.py > $R/gpurun_out/${TAG}_bench_default.json 2> $R/gpurun_out/${TAG}_bench_default.err
bash $R/tools/profile_round.sh $TAG
bash $R/tools/pmc_sq.sh $TAG
cd $R
python3 tools/pmc_classes.py gpurun_out/prof_${TAG}_pmc_fetch_size.txt gpurun_out/prof_${TAG}_pmc_write_size.txt > gpurun_out/${TAG}_pmc_per_class.json
python3 tools/sq_summary.py gpurun_out/pmc_${TAG}_1.txt gpurun_out/pmc_${TAG}_2.txt > gpurun_out/${TAG}_sq_counters.txt
python3 tools/mfma_summary.py gpurun_out/pmc_${TAG}_3.txt gpurun_out/prof_${TAG}_kernel_trace.txt --json gpurun_out/${TAG}_mfma_per_class.json > gpurun_out/${TAG}_mfma_busy.txt
MNAS_BENCH_DETAIL=1 python3 bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench_detail.json 2> gpurun_out/${TAG}_detail.txt
python3 bench.py --h2d --no-cpu-baseline --no-roofline > gpurun_out/${TAG}_bench_h2d.json 2>/dev/null
python3 bench.py --k5 --no-cpu-baseline > gpurun_out/${TAG}_bench_k5.json 2>/dev/null
python3 bench.py --se --no-cpu-baseline > gpurun_out/${TAG}_bench_se.json 2>/dev/null
for hw in 384x512 512x512 512x384; do
  python3 bench.py --hw $hw --batch 64 --no-cpu-baseline --no-roofline > gpurun_out/${TAG}_bench_${hw}.json 2>/dev/null
done
python3 bench.py --clusters --no-cpu-baseline > gpurun_out/${TAG}_bench_clusters.json 2>/dev/null
python3 bench.py --ccf --no-cpu-baseline > gpurun_out/${TAG}_bench_ccf.json 2>/dev/null
