"""Per-layer micro-benchmark of the 1x1 forward (mnas_conv_gemm mode 0) at the bench shapes (bs 256): us per launch and
algorithmic TB/s, with a torch.matmul check of the full-size result.  Buffers rotate over several sets so that nothing
stays in the 256 MiB Infinity Cache between iterations (an in-situ number is still the final word: MNAS_BENCH_DETAIL=1
python bench.py).  A/B across kernels with the MNAS_PWF=0/1 environment switch (one process per setting)."""
import ctypes as C, sys, os, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from mnasnet_pytorch_amd import _lib as L
lib = L.load()
SHAPES = [(112, 32, 16), (112, 16, 48), (112, 48, 16), (56, 24, 72), (56, 72, 24), (28, 40, 240), (28, 240, 40),
          (14, 80, 480), (14, 480, 80), (14, 96, 576), (14, 576, 96), (7, 192, 1152), (7, 1152, 192)]
N = int(os.environ.get("KB_N", "256"))
def run(H, Ci, Co, check=True):
    M = N * H * H
    nset = max(2, min(8, int(600e6 / (M * (Ci + Co) * 2)) + 1))
    bf = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
    xs = [bf(M, Ci) for _ in range(nset)]
    ys = [torch.empty(M, Co, dtype=torch.bfloat16, device="cuda") for _ in range(nset)]
    sc, sh = torch.rand(Ci, device="cuda") + 0.5, torch.randn(Ci, device="cuda") * 0.2
    w = torch.randn(Co, Ci, 1, 1, device="cuda") * (1.0 / Ci ** 0.5)
    bias = torch.randn(Co, device="cuda") * 0.1
    wp = torch.empty(lib.mnas_packed_bytes(L.PACK_FWD, Co, Ci, 1, 1), dtype=torch.uint8, device="cuda")
    L.check(lib.mnas_pack_weights(w.data_ptr(), L.PACK_FWD, Co, Ci, 1, 1, wp.data_ptr(), L.cur_stream()))
    nparts = lib.mnas_conv_gemm_parts(0, M, Ci, Co, 1)
    if nparts < 1:
        tp = 128 if M >= 20000 else lib.mnas_conv_gemm_tile_pixels(M, Co, Ci)
        nparts = max(1, min(1024, (M + tp - 1) // tp))
    stats = torch.full((2, Co, nparts), float("nan"), device="cuda")
    def call(i):
        a = L.MnasConvGemm()
        a.mode, a.N, a.Hi, a.Wi, a.Ci, a.Ho, a.Wo, a.Co = 0, N, H, H, Ci, H, H, Co
        a.kh = a.kw = 1; a.stride, a.pad, a.nparts = 1, 0, nparts
        a.act = L.MnasActIn(xs[i].data_ptr(), sc.data_ptr(), sh.data_ptr()) if not os.environ.get("KB_NOCOEF") else L.MnasActIn(xs[i].data_ptr(), None, None)
        a.w, a.bias, a.out, a.stats = wp.data_ptr(), bias.data_ptr(), ys[i].data_ptr(), stats.data_ptr()
        L.check(lib.mnas_conv_gemm(C.byref(a), L.cur_stream()), "conv_gemm")
    for i in range(nset): call(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 3 * nset
    e0.record()
    for i in range(iters): call(i % nset)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    err = serr = float("nan")
    if check:
        a = torch.relu(xs[0].float() * sc + sh).to(torch.bfloat16).float() if not os.environ.get("KB_NOCOEF") else xs[0].float()
        ref = a @ w.view(Co, Ci).to(torch.bfloat16).float().t() + bias
        call(0); torch.cuda.synchronize()
        err = float((ys[0].float() - ref).abs().max() / ref.abs().max())
        s1 = stats[0].double().sum(-1); s2 = stats[1].double().sum(-1)
        serr = max(float((s1 - ref.double().sum(0)).abs().max() / ref.double().sum(0).abs().max()),
                   float((s2 - (ref.double() ** 2).sum(0)).abs().max() / (ref.double() ** 2).sum(0).abs().max()))
    print("pw fwd H=%3d %4d->%4d nparts=%4d : %7.1f us  %5.2f TB/s   maxerr %.2e  stats err %.2e" %
          (H, Ci, Co, nparts, us, M * (Ci + Co) * 2 / us / 1e6, err, serr), flush=True)
sel = os.environ.get("KB_SHAPES")
for s in SHAPES:
    if sel and ("%d-%d" % (s[1], s[2])) not in sel.split(","):
        continue
    run(*s, check=not os.environ.get("KB_NOCHECK"))
