#!/bin/bash
# Build an earlier commit of THIS repo next to HEAD so that one gpurun call can run both on the same box:
#     bash tools/repro/make_tree.sh d320a78 r03      -> tools/repro/r03/{bench.py, mnasnet_pytorch_amd/, oracle/, include/, profiles/}
# (git-ignored; built .so included, objects removed).  tools/repro/run.sh then runs HEAD / that tree / HEAD with the driver's command.
set -e
REV=${1:?commit}; TAG=${2:?tag}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REV" | tar -x -C "$TMP"
make -C "$TMP/mnasnet_pytorch_amd/csrc" -j6 > "$TMP/build.log" 2>&1
D="$ROOT/tools/repro/$TAG"
rm -rf "$D"; mkdir -p "$D/profiles"
cp -r "$TMP/bench.py" "$TMP/mnasnet_pytorch_amd" "$TMP/oracle" "$TMP/include" "$D/"
cp "$TMP"/profiles/*per_class.json "$D/profiles/" 2>/dev/null || true
find "$D" -name "*.o" -delete
echo "$REV" > "$D/REV"
rm -rf "$TMP"
echo "built $D"
