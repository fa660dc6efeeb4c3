#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in alt b; do
  MNAS_LIB_PATH=$PWD/mnasnet_pytorch_amd/csrc/libmnas_hip_$v.so MNAS_BENCH_DETAIL=1 python3 bench.py --no-cpu-baseline > gpurun_out/o_v.json 2> gpurun_out/o_detail_$v.txt
  python3 -c "
import json; r=json.load(open('gpurun_out/o_v.json')); print('LIB=$v', r['value'], r['ms_per_step'], r['kernel_classes']['k_wgrad']['ms_per_step'], r['kernel_classes']['k_igemm<dgrad>']['ms_per_step'])"
  grep "k_wgrad \|k_igemm<dgrad>" gpurun_out/o_detail_$v.txt | awk '{print $1, $2, $3}' | grep ",1,1" | sort | uniq -c | awk '{print "   ", $0}'
done
