cd $GRAFT_REPO_ROOT
L=$PWD/mnasnet_pytorch_amd/csrc
export MNAS_LIB_PATH=$L/libmnas_hip_bw.so
echo "== kernel tests BW=2"; MNAS_DW_BW=2 python3 -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "dw_fwd" 2>&1 | tail -3
for r in 1 2 3; do for bw in 4 2; do echo "== fwd BW=$bw"; MNAS_DW_BW=$bw python3 tools/kbench_dw.py fwd 2>/dev/null | grep "^dw" | cut -c1-60; done; done
