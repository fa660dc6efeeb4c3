"""What a tiny dependent launch costs on this box: a stream of empty kernels (1 / 64 / 1024 blocks) against the BatchNorm forward
finalize at the shapes of the step (48 channels x 1024 partial columns; 576 x 170).  us per launch from HIP events over 2000 launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mnasnet_pytorch_amd import _lib as L
lib = L.load()


def timeit(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


s = L.cur_stream()
for blocks, threads in ((1, 64), (1, 256), (64, 256), (1024, 256)):
    print("empty kernel %4d x %3d : %5.2f us per launch" % (blocks, threads, timeit(lambda: lib.mnas_probe_empty(blocks, threads, s))))
for C_, P in ((48, 1024), (576, 170), (16, 2048), (1152, 98)):
    part = torch.rand(2, C_, P, device="cuda")
    g, b = torch.ones(C_, device="cuda"), torch.zeros(C_, device="cuda")
    rm, rv = torch.zeros(C_, device="cuda"), torch.ones(C_, device="cuda")
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    bn = torch.zeros(8, C_, device="cuda")
    f = lambda: lib.mnas_bn_fwd_finalize(part.data_ptr(), P, C_, 1e6, g.data_ptr(), b.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                         nbt.data_ptr(), 0.1, 1e-5, 1, bn.data_ptr(), s)
    print("bn_fwd_finalize C=%4d P=%4d : %5.2f us per launch" % (C_, P, timeit(f)))
