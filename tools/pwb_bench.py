"""Micro-benchmark of mnas_pw_bwd at bench sizes (prints us per launch).  CAUTION: it re-runs one launch on the same
buffers, so up to 256 MB of them stay in the Infinity Cache between iterations -- the numbers are 20-40 % better than
in the training step (use MNAS_NO_SIDE=1 MNAS_BENCH_DETAIL=1 python bench.py for in-situ times).  Good for A/B only."""
import ctypes as C, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from mnasnet_pytorch_amd import _lib as L
lib = L.load()
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (H, Ci, Co) in [(112, 48, 16), (112, 16, 48), (112, 32, 16), (56, 72, 24), (56, 24, 72), (28, 240, 40), (28, 40, 240)]:
    N = 256; M = N * H * H
    bf = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
    x, g, y, resid = bf(M, Ci), bf(M, Co), bf(M, Co), bf(M, Ci)
    bn_o = torch.rand(8, Co, device="cuda") + 0.5; bn_i = torch.rand(8, Ci, device="cuda") + 0.5
    w = torch.randn(Co, Ci, 1, 1, device="cuda") * 0.1
    wp = torch.empty(lib.mnas_packed_bytes(L.PACK_DGRAD, Co, Ci, 1, 1), dtype=torch.uint8, device="cuda")
    L.check(lib.mnas_pack_weights(w.data_ptr(), L.PACK_DGRAD, Co, Ci, 1, 1, wp.data_ptr(), L.cur_stream()))
    gin = torch.empty(M, Ci, dtype=torch.bfloat16, device="cuda")
    nparts = 1024 if M >= 800000 else 512
    wpart = torch.empty(nparts, Co, Ci, device="cuda"); redp = torch.empty(2, Ci, nparts, device="cuda")
    c = L.MnasPwBwd(); c.M, c.Ci, c.Co, c.nparts = M, Ci, Co, nparts
    c.x = L.MnasActIn(x.data_ptr(), bn_i.data_ptr(), bn_i.data_ptr() + 4 * Ci)
    c.dy = L.MnasGradIn(g.data_ptr(), y.data_ptr(), bn_o.data_ptr())
    c.w, c.gin, c.wpartial = wp.data_ptr(), gin.data_ptr(), wpart.data_ptr()
    c.red_partial, c.red_y, c.red_bn = redp.data_ptr(), x.data_ptr(), bn_i.data_ptr()
    t = bench(lambda: L.check(lib.mnas_pw_bwd(C.byref(c), L.cur_stream())))
    units = 2 * Co + 2 * Ci
    print("pw_bwd H=%3d Ci=%3d Co=%3d : %7.1f us  (%5.2f TB/s on g,y,x,gin)" % (H, Ci, Co, t, M * units * 2 / t / 1e6))
