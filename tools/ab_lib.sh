#!/bin/bash
# A/B the in-tree library against libmnas_hip_alt.so (tools/build_alt.sh) over bench.py in ONE gpurun call
for v in main alt; do
  if [ $v = alt ]; then export MNAS_LIB_PATH=$PWD/mnasnet_pytorch_amd/csrc/libmnas_hip_alt.so; else unset MNAS_LIB_PATH; fi
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ablib_$v.txt 2> gpurun_out/ablib_$v.err
  python3 - "$v" gpurun_out/ablib_$v.txt <<'PY'
import sys, json
for l in open(sys.argv[2]):
    if l.startswith('{'):
        d = json.loads(l)
        kc = d.get('kernel_classes', {})
        print(sys.argv[1], d['value'], d['ms_per_step'], ' '.join('%s=%.3f' % (k.replace('k_', ''), v['ms_per_step']) for k, v in kc.items()))
PY
done
