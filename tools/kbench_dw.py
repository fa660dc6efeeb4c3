"""Stand-alone micro-benchmark of the depthwise sweeps (mnas_dw_fwd / mnas_dw_bwd phase 0) at the bench shapes (bs 256): us per
launch and algorithmic TB/s (forward 2 tensors, fused backward 4 tensors, bf16), buffers rotating over several sets so that
nothing stays in the 256 MiB Infinity Cache between launches.  A/B two builds with MNAS_LIB_PATH (one process per library).
    python3 tools/kbench_dw.py [fwd|bwd|all] [shape filter, e.g. 14x576x5]        KB_N=256  KB_PARTS=<nparts override>"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mnasnet_pytorch_amd import _lib as L
lib = L.load()
N = int(os.environ.get("KB_N", "256"))
# (H=W, C, k): MNASNet-1.0 (ccf=False) depthwise layers + the --k5 variant's + ccf=True's 5x5 layers
SHAPES = [(112, 32, 3), (112, 48, 3), (56, 72, 5), (28, 240, 5), (14, 480, 3), (14, 576, 5), (7, 1152, 3),
          (112, 48, 5), (14, 480, 5), (7, 1152, 5), (28, 120, 5)]
which = sys.argv[1] if len(sys.argv) > 1 else "all"
filt = sys.argv[2] if len(sys.argv) > 2 else ""


def bf(*s):
    return (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)


def timeit(call, nset):
    for i in range(nset):
        call(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = max(6, 3 * nset)
    e0.record()
    for i in range(iters):
        call(i % nset)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def run(H, Cc, k):
    M = N * H * H
    e = M * Cc
    nset = max(2, min(8, int(700e6 / (e * 2 * 4)) + 1))
    w = torch.randn(k * k, Cc, device="cuda") * 0.2
    bias = torch.randn(Cc, device="cuda") * 0.1
    sc, sh = torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda") * 0.2
    coef = torch.randn(8, Cc, device="cuda") * 0.3
    coef[0], coef[1] = sc, sh
    coef[6] = coef[6].abs() + 0.5
    xs = [bf(N, H, H, Cc) for _ in range(nset)]
    res = {}
    if which in ("fwd", "all"):
        outs = [torch.empty(N, H, H, Cc, dtype=torch.bfloat16, device="cuda") for _ in range(nset)]
        nl = int(os.environ.get("KB_PARTS", "0")) or max(64, min(2048, (e + 8191) // 8192))
        rows = lib.mnas_dw_rows(N, H, H, Cc, k, nl, 0)
        stats = torch.empty(2 * Cc * max(rows, 1), device="cuda")

        def fwd(i):
            a = L.MnasDwFwd()
            a.N, a.H, a.W, a.C, a.k, a.nparts = N, H, H, Cc, k, nl
            a.in_ = L.MnasActIn(xs[i].data_ptr(), sc.data_ptr(), sh.data_ptr())
            a.w, a.bias, a.out, a.stats = w.data_ptr(), bias.data_ptr(), outs[i].data_ptr(), stats.data_ptr()
            L.check(lib.mnas_dw_fwd(C.byref(a), L.cur_stream()), "dw_fwd")
        us = timeit(fwd, nset)
        res["fwd"] = (us, 2 * 2 * e / us / 1e6)
        del outs
    if which in ("bwd", "all"):
        gs = [bf(N, H, H, Cc) for _ in range(nset)]
        ys = [bf(N, H, H, Cc) for _ in range(nset)]
        gins = [torch.empty(N, H, H, Cc, dtype=torch.bfloat16, device="cuda") for _ in range(nset)]
        nl = int(os.environ.get("KB_PARTS", "0")) or max(64, min(1024, (e + 8191) // 8192))
        rows = lib.mnas_dw_rows(N, H, H, Cc, k, nl, 1)
        wpart = torch.empty(max(rows, 1) * k * k * Cc, device="cuda")
        red = torch.empty(2 * Cc * max(rows, 1), device="cuda")
        for gm in (0, 1):
            def bwd(i):
                a = L.MnasDwBwd()
                a.N, a.H, a.W, a.C, a.k, a.nparts = N, H, H, Cc, k, nl
                a.x = L.MnasActIn(xs[i].data_ptr(), sc.data_ptr(), sh.data_ptr())
                a.dy = L.MnasGradIn(gs[i].data_ptr(), ys[i].data_ptr(), coef.data_ptr())
                a.w, a.gin, a.wpartial = w.data_ptr(), gins[i].data_ptr(), wpart.data_ptr()
                a.red_bn, a.red_partial, a.phase, a.g_masked = coef.data_ptr(), red.data_ptr(), 0, gm
                L.check(lib.mnas_dw_bwd(C.byref(a), L.cur_stream()), "dw_bwd")
            us = timeit(bwd, nset)
            res["bwd_gm%d" % gm] = (us, 2 * 4 * e / us / 1e6)
    geo = (C.c_int * 7)()
    lib.mnas_dw_geometry(N, H, H, Cc, k, 1, geo)
    print("dw %3dx%-3d C=%4d k=%d  " % (H, H, Cc, k) + "  ".join("%s %7.1f us %5.2f TB/s" % (kk, v[0], v[1]) for kk, v in res.items())
          + "   [bwd geometry: cpw %d sx %d thr %d strips %d cblocks %d lds %d G %d]" % tuple(geo), flush=True)


for s in SHAPES:
    if filt and ("%dx%dx%d" % s) not in filt.split(","):
        continue
    run(*s)
