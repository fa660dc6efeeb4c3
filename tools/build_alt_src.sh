#!/bin/bash
# A/B against an OLDER VERSION of one source file (e.g. the last commit's): builds the diagnosis library with <name>.hip replaced by
# the given file.      bash tools/build_alt_src.sh mnas_dw.hip <(git show HEAD~1:mnasnet_pytorch_amd/csrc/mnas_dw.hip) libmnas_hip_old.so
# (the other objects come from / go to the same cache as tools/build_alt.sh)
set -e
NAME=$1; ALT=$2; OUT=${3:-libmnas_hip_old.so}
cd "$(dirname "$0")/../mnasnet_pytorch_amd/csrc"
OBJ=/tmp/mnas_diag_obj
mkdir -p $OBJ
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -fno-slp-vectorize -DMNAS_DIAG -I$PWD"
cat "$ALT" > $OBJ/old_$NAME
LINK=""
for f in *.hip; do
    o=$OBJ/${f%.hip}.o
    SCHED="-mllvm -amdgpu-sched-strategy=max-ilp"            # as the Makefile: every file but the depthwise sweeps
    [ "$f" = "mnas_dw.hip" ] && SCHED=""
    if [ "$f" = "$NAME" ]; then
        o=$OBJ/${f%.hip}.old.o
        /opt/rocm/bin/hipcc $FLAGS $SCHED -c $OBJ/old_$NAME -o $o &
    elif [ ! -f $o ] || [ $f -nt $o ] || [ mnas_common.h -nt $o ] || [ ../../include/mnas.h -nt $o ] || [ Makefile -nt $o ]; then
        /opt/rocm/bin/hipcc $FLAGS $SCHED -c $f -o $o &
    fi
    LINK="$LINK $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $LINK -o $OUT
echo built $OUT
