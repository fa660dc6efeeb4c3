#!/bin/bash
# Whole-step sweep of the engine's grid knobs (bench.py overrides), one line per setting, the default interleaved:  bash tools/sweep_knobs.sh
cd ${GRAFT_REPO_ROOT:-$PWD}
run() {  # name, env assignments...
  name=$1; shift
  env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-box > gpurun_out/knob.txt 2>/dev/null
  python3 - "$name" <<'PY'
import sys, json
d = json.loads([l for l in open("gpurun_out/knob.txt") if l.startswith("{")][-1])
kc = d["kernel_classes"]
print("%-28s %9.1f img/s %7.3f ms  " % (sys.argv[1], d["value"], d["ms_per_step"]) + " ".join("%s=%.3f" % (k.replace("k_", ""), v["ms_per_step"]) for k, v in list(kc.items())[:6]), flush=True)
PY
}
run default MNAS_X=0
for v in 768 1280 2048; do run DWB_PARTS=$v MNAS_DWB_PARTS=$v; done
run default MNAS_X=0
for v in 768,512,80 1536,512,80 1024,384,80 1024,768,80 1024,512,64 1024,512,96 1024,512,128; do run PWB=$v MNAS_PWB=$v; done
run default MNAS_X=0
for v in 256 768 1024; do run WGRAD_WGS=$v MNAS_WGRAD_WGS=$v; done
for v in 256 768 1024; do run PWB_SEGMENTS=$v MNAS_PWB_SEGMENTS=$v; done
run default MNAS_X=0
