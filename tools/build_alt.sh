#!/bin/bash
# Build the DIAGNOSIS library: every source with -DMNAS_DIAG (the MNAS_* environment switches documented in DESIGN.md exist only
# in this build; the shipped libmnas_hip.so reads no environment), optionally one source with extra defines for an in-one-call A/B:
#     bash tools/build_alt.sh                       # plain diagnosis build
#     bash tools/build_alt.sh mnas_dw.hip -DMNAS_DW_XFILL=0
#     OUT=libmnas_hip_b.so bash tools/build_alt.sh mnas_dw.hip -DFOO=1     # a second variant next to the first
#     EXTRA="-DMNAS_UWAVE=0" OUT=libmnas_hip_c.so bash tools/build_alt.sh  # defines for EVERY source (own object cache per EXTRA string)
# -> mnasnet_pytorch_amd/csrc/libmnas_hip_alt.so (git-ignored); run with MNAS_LIB_PATH=$PWD/mnasnet_pytorch_amd/csrc/libmnas_hip_alt.so
# The object built with extra defines goes to <name>.alt.o and is linked INSTEAD of the plain one: a later plain build never
# picks up a stale variant object.  Plain objects are rebuilt when the source, mnas_common.h or include/mnas.h is newer.
set -e
cd "$(dirname "$0")/../mnasnet_pytorch_amd/csrc"
OBJ=/tmp/mnas_diag_obj
[ -n "$EXTRA" ] && OBJ=/tmp/mnas_diag_obj_$(echo "$EXTRA" | md5sum | cut -c1-8)
OUT=${OUT:-libmnas_hip_alt.so}
mkdir -p $OBJ
SRC=${1:-}; [ $# -gt 0 ] && shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -fno-slp-vectorize -DMNAS_DIAG $EXTRA"
LINK=""
for f in *.hip; do
    o=$OBJ/${f%.hip}.o
    SCHED="-mllvm -amdgpu-sched-strategy=max-ilp"            # as the Makefile: every file but the depthwise sweeps
    [ "$f" = "mnas_dw.hip" ] && SCHED=""
    if [ "$f" = "$SRC" ] && [ $# -gt 0 ]; then
        o=$OBJ/${f%.hip}.alt.o
        /opt/rocm/bin/hipcc $FLAGS $SCHED "$@" -c $f -o $o &
    elif [ ! -f $o ] || [ $f -nt $o ] || [ mnas_common.h -nt $o ] || [ ../../include/mnas.h -nt $o ] || [ Makefile -nt $o ]; then
        /opt/rocm/bin/hipcc $FLAGS $SCHED -c $f -o $o &
    fi
    LINK="$LINK $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $LINK -o $OUT
echo built $OUT
