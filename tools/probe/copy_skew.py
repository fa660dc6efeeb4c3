"""Does the relative alignment of a copy's source and destination matter on this HBM?  mnas_probe_copy (plain, one load in flight)
and mnas_probe_copy4 (nontemporal, four in flight) with the destination skewed by a few strides against a 2 MiB-aligned source.
    python3 tools/probe/copy_skew.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mnasnet_pytorch_amd import _lib as L
lib = L.load()
nbytes = 1 << 30
pad = 1 << 22
src = torch.empty(nbytes + pad, dtype=torch.uint8, device="cuda").fill_(1)
dst = torch.empty(nbytes + pad, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream


def align(t):
    p = t.data_ptr()
    return (p + (1 << 21) - 1) & ~((1 << 21) - 1)


def rate(four, sp, dp):
    best = 0.0
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if four:
            lib.mnas_probe_copy4(sp, dp, nbytes, 2048, s)
        else:
            lib.mnas_probe_copy(sp, dp, nbytes, s)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 2 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    return best


sp, dp = align(src), align(dst)
for skew in (0, 256, 1024, 4096, 4352, 65536 + 256, (1 << 20) + 4352):
    print("skew %8d B: plain copy %7.1f GB/s   nontemporal four-deep %7.1f GB/s" %
          (skew, rate(False, sp, dp + skew), rate(True, sp, dp + skew)), flush=True)
