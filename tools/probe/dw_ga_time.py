"""Timing probe: the fused depthwise backward with and without the on-read affine map of g (MnasDwBwd.g_gate / g_bias) at a bench
shape, same buffers, rotating sets so that nothing stays in the Infinity Cache.  usage: python3 tools/probe/dw_ga_time.py [H C k]"""
import ctypes as C, sys, os, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from mnasnet_pytorch_amd import _lib as L
lib = L.load()
H, Cc, k = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (112, 48, 5)
N, nparts = 256, 1024
bf = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
nset = 3
xs, gs, ys = [bf(N, H, H, Cc) for _ in range(nset)], [bf(N, H, H, Cc) for _ in range(nset)], [bf(N, H, H, Cc) for _ in range(nset)]
gin = torch.empty((N, H, H, Cc), dtype=torch.bfloat16, device="cuda")
w = torch.randn(k * k, Cc, device="cuda") * 0.1
sc, sh = torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda") * 0.2
coef = torch.randn(8, Cc, device="cuda"); coef[6] = coef[6].abs() + 0.5
bnin = torch.randn(8, Cc, device="cuda"); bnin[0], bnin[1] = sc, sh
rows = lib.mnas_dw_rows(N, H, H, Cc, k, nparts, 1)
wpart = torch.empty((rows, k * k, Cc), device="cuda")
redp = torch.empty((2, Cc, rows), device="cuda")
gate, bias = torch.rand(N, Cc, device="cuda"), torch.randn(N, Cc, device="cuda") * 0.01
def call(i, ga):
    a = L.MnasDwBwd()
    a.N, a.H, a.W, a.C, a.k, a.nparts, a.phase = N, H, H, Cc, k, nparts, 0
    a.x = L.MnasActIn(xs[i].data_ptr(), sc.data_ptr(), sh.data_ptr())
    a.dy = L.MnasGradIn(gs[i].data_ptr(), ys[i].data_ptr(), coef.data_ptr())
    a.w, a.gin, a.wpartial = w.data_ptr(), gin.data_ptr(), wpart.data_ptr()
    a.red_bn, a.red_partial = bnin.data_ptr(), redp.data_ptr()
    if ga:
        a.g_gate, a.g_bias = gate.data_ptr(), bias.data_ptr()
    L.check(lib.mnas_dw_bwd(C.byref(a), L.cur_stream()), "dw_bwd")
for ga in (0, 1, 0, 1):
    for i in range(nset): call(i, ga)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(4 * nset): call(i % nset, ga)
    e1.record(); torch.cuda.synchronize()
    print("H=%d C=%d k=%d  g affine on read=%d : %.1f us" % (H, Cc, k, ga, e0.elapsed_time(e1) * 1e3 / (4 * nset)))
