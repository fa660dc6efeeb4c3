#!/bin/bash
# Build the DIAGNOSIS library: every source with -DMNAS_DIAG (the MNAS_* environment switches documented in DESIGN.md exist only
# in this build; the shipped libmnas_hip.so reads no environment), optionally one source with extra defines for an in-one-call A/B:
#     bash tools/build_alt.sh                       # plain diagnosis build
#     bash tools/build_alt.sh mnas_dw.hip -DMNAS_DW_XFILL=0
# -> mnasnet_pytorch_amd/csrc/libmnas_hip_alt.so (git-ignored); run with MNAS_LIB_PATH=$PWD/mnasnet_pytorch_amd/csrc/libmnas_hip_alt.so
set -e
cd "$(dirname "$0")/../mnasnet_pytorch_amd/csrc"
OBJ=/tmp/mnas_diag_obj
mkdir -p $OBJ
SRC=${1:-}; [ $# -gt 0 ] && shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -fno-slp-vectorize -DMNAS_DIAG"
for f in *.hip; do
    o=$OBJ/${f%.hip}.o
    if [ "$f" = "$SRC" ]; then
        /opt/rocm/bin/hipcc $FLAGS "$@" -c $f -o $o &
    elif [ ! -f $o ] || [ $f -nt $o ] || [ mnas_common.h -nt $o ]; then
        /opt/rocm/bin/hipcc $FLAGS -c $f -o $o &
    fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o libmnas_hip_alt.so
echo built libmnas_hip_alt.so
