#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in alt b alt b; do
  MNAS_LIB_PATH=$PWD/mnasnet_pytorch_amd/csrc/libmnas_hip_$v.so MNAS_BENCH_DETAIL=1 python3 bench.py --no-cpu-baseline > gpurun_out/o_v.json 2> gpurun_out/o_detail_$v.txt
  python3 -c "
import json; r=json.load(open('gpurun_out/o_v.json')); print('LIB=$v', r['value'], r['ms_per_step'], r['kernel_classes']['k_igemm<fwd>']['ms_per_step'], r['kernel_classes']['k_igemm<dgrad>']['ms_per_step'])"
done
