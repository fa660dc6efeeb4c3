#!/bin/bash
# Build a second library with extra compiler defines for an in-one-call A/B:  bash tools/build_alt.sh mnas_dw.hip -DMNAS_DW_XFILL=0
# -> mnasnet_pytorch_amd/csrc/libmnas_hip_alt.so (git-ignored); run with MNAS_LIB_PATH=$PWD/mnasnet_pytorch_amd/csrc/libmnas_hip_alt.so
set -e
cd "$(dirname "$0")/../mnasnet_pytorch_amd/csrc"
SRC=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -fno-slp-vectorize "$@" -c $SRC -o /tmp/alt_${SRC%.hip}.o
OBJS=$(ls *.o | grep -v "^${SRC%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/alt_${SRC%.hip}.o -o libmnas_hip_alt.so
echo built libmnas_hip_alt.so
