#!/bin/bash
# Same-call A/B of several builds of the library over bench.py, interleaved and repeated:  bash tools/ab_libs.sh <reps> <name|main> ...
# (name -> mnasnet_pytorch_amd/csrc/libmnas_hip_<name>.so, "main" = the shipped library).  One summary line per run.
REPS=$1; shift
L=$PWD/mnasnet_pytorch_amd/csrc
for r in $(seq 1 $REPS); do for v in "$@"; do
  if [ $v = main ]; then unset MNAS_LIB_PATH; else export MNAS_LIB_PATH=$L/libmnas_hip_$v.so; fi
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-box > gpurun_out/ablibs_$v.txt 2> gpurun_out/ablibs_$v.err
  python3 - "$v" gpurun_out/ablibs_$v.txt <<'PY'
import sys, json
for l in open(sys.argv[2]):
    if l.startswith('{'):
        d = json.loads(l)
        kc = d.get('kernel_classes', {})
        print("%-8s %9.1f img/s %7.3f ms  " % (sys.argv[1], d['value'], d['ms_per_step']) + ' '.join('%s=%.3f' % (k.replace('k_', ''), v['ms_per_step']) for k, v in list(kc.items())[:8]), flush=True)
PY
done; done
