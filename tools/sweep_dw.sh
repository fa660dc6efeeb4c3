#!/bin/bash
# Geometry sweep of the depthwise sweeps with the diagnosis build (MNAS_DW_CPW / MNAS_DW_SX / MNAS_DW_G, KB_PARTS): one line per point.
#   bash tools/sweep_dw.sh <fwd|bwd> <HxCxk> "<cpw list>" "<sx list>" "<G list>" "<parts list>"
MODE=$1; SHAPE=$2; CPWS=$3; SXS=$4; GS=$5; PARTS=${6:-0}
R=${GRAFT_REPO_ROOT:-$PWD}
export MNAS_LIB_PATH=$R/mnasnet_pytorch_amd/csrc/libmnas_hip_alt.so
echo "== default"; python3 $R/tools/kbench_dw.py $MODE $SHAPE 2>/dev/null | grep "^dw"
for P in $PARTS; do for C in $CPWS; do for S in $SXS; do for G in $GS; do
  OUT=$(KB_PARTS=$P MNAS_DW_CPW=$C MNAS_DW_SX=$S MNAS_DW_G=$G python3 $R/tools/kbench_dw.py $MODE $SHAPE 2>/dev/null | grep "^dw")
  echo "parts=$P cpw=$C sx=$S G=$G :: $OUT"
done; done; done; done
