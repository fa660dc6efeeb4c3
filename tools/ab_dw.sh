#!/bin/bash
# A/B of the depthwise sweeps in ONE gpurun call: tools/kbench_dw.py for the in-tree library and for the diagnosis build
# (tools/build_alt.sh -> libmnas_hip_alt.so) under a list of environment settings.   bash tools/ab_dw.sh <fwd|bwd|all> "<shapes>" "VAR=1 VAR2=3" "VAR=0" ...
MODE=${1:-fwd}; SHAPES=${2:-}; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
echo "== main library"; python3 $R/tools/kbench_dw.py $MODE "$SHAPES"
for SET in "$@"; do
  echo "== alt library: $SET"
  env MNAS_LIB_PATH=$R/mnasnet_pytorch_amd/csrc/libmnas_hip_alt.so $SET python3 $R/tools/kbench_dw.py $MODE "$SHAPES"
done
