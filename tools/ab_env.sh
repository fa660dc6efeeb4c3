#!/bin/bash
# A/B a diagnosis environment variable over bench.py in ONE gpurun call:  bash tools/ab_env.sh VAR v1 v2 ...
# writes gpurun_out/ab_<VAR>_<v>.txt and prints one summary line per value (img/s, ms/step, per-class ms)
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_${VAR}_${v}.txt 2> gpurun_out/ab_${VAR}_${v}.err
  python3 - "$VAR=$v" gpurun_out/ab_${VAR}_${v}.txt <<'PY'
import sys, json
for l in open(sys.argv[2]):
    if l.startswith('{'):
        d = json.loads(l)
        kc = d.get('kernel_classes', {})
        print(sys.argv[1], d['value'], d['ms_per_step'], ' '.join('%s=%.3f' % (k.replace('k_', ''), v['ms_per_step']) for k, v in kc.items()))
PY
done
