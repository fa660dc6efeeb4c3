#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 0 512 0 512; do
  MNAS_PWB_SEGMENTS=$v python3 bench.py --no-cpu-baseline > gpurun_out/o_v.json 2>/dev/null
  python3 -c "
import json; r=json.load(open('gpurun_out/o_v.json')); print('SEG=$v', r['value'], r['ms_per_step'], r['kernel_classes']['k_pw_bwd']['ms_per_step'])"
done
timeout 900 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_train.py -q -x 2>&1 | grep -E "passed|failed" | tail -3
