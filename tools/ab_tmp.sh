#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_se.py -q 2>&1 | tail -6
for v in 0 1; do
  MNAS_NO_SE_ONLOAD=$v MNAS_BENCH_DETAIL=1 python3 bench.py --se --no-cpu-baseline > gpurun_out/o_se$v.json 2> gpurun_out/o_se_detail$v.txt
  python3 -c "
import json; r=json.load(open('gpurun_out/o_se$v.json')); print('NO_ONLOAD=$v', r['value'], r['ms_per_step'])
for a,b in sorted(r['kernel_classes'].items(), key=lambda t:-t[1]['ms_per_step'])[:9]: print('    ',a,b['ms_per_step'],b['launches'])"
done
