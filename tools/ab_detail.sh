#!/bin/bash
# A/B of the in-tree library against libmnas_hip_alt.so with the per-launch table (MNAS_BENCH_DETAIL) of one kernel class:
#     bash tools/ab_detail.sh k_pw_bwd [bench flags]      -> gpurun_out/abd_{main,alt}.{txt,err}; prints both tables side by side
KEY=${1:-k_pw_bwd}; shift
for v in main alt main alt; do
  if [ $v = alt ]; then export MNAS_LIB_PATH=$PWD/mnasnet_pytorch_amd/csrc/libmnas_hip_alt.so; else unset MNAS_LIB_PATH; fi
  MNAS_BENCH_DETAIL=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-box "$@" > gpurun_out/abd_$v.txt 2> gpurun_out/abd_$v.err
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/abd_$v.txt | head -1 | sed "s/^/$v /"
done
paste <(grep "^$KEY" gpurun_out/abd_main.err) <(grep "^$KEY" gpurun_out/abd_alt.err | awk '{print $(NF-3), $(NF-2), $(NF-1), $NF}')
