"""Per-layer micro-benchmark of the 1x1 input gradient (mnas_conv_gemm mode 1, dy-on-load + fused BatchNorm-backward reduce)
at the bench shapes of the narrowing (project) convs, bs 256.  A/B: MNAS_PWD=0/1 (k_igemm vs the DMA-pipelined kernel)."""
import ctypes as C, sys, os, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from mnasnet_pytorch_amd import _lib as L
lib = L.load()
SHAPES = [(7, 1152, 192), (112, 48, 16), (56, 72, 24), (28, 240, 40), (28, 120, 40), (14, 480, 80), (14, 480, 96), (14, 576, 96)]   # H, conv Ci, conv Co
N = int(os.environ.get("KB_N", "256"))
def run(H, Ci, Co):
    M = N * H * H
    nset = max(2, min(8, int(600e6 / (M * (2 * Ci + 2 * Co) * 2)) + 1))
    bf = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
    gs, ys = [bf(M, Co) for _ in range(nset)], [bf(M, Co) for _ in range(nset)]
    rys = [bf(M, Ci) for _ in range(nset)]
    outs = [torch.empty(M, Ci, dtype=torch.bfloat16, device="cuda") for _ in range(nset)]
    coef = torch.rand(8, Co, device="cuda") + 0.5
    rbn = torch.rand(8, Ci, device="cuda") + 0.5
    w = torch.randn(Co, Ci, 1, 1, device="cuda") * (1.0 / Ci ** 0.5)
    wp = torch.empty(lib.mnas_packed_bytes(L.PACK_DGRAD, Co, Ci, 1, 1), dtype=torch.uint8, device="cuda")
    L.check(lib.mnas_pack_weights(w.data_ptr(), L.PACK_DGRAD, Co, Ci, 1, 1, wp.data_ptr(), L.cur_stream()))
    nparts = lib.mnas_conv_gemm_parts(1, M, Co, Ci, 1)
    if nparts < 1:
        tp = 128 if M >= 20000 else lib.mnas_conv_gemm_tile_pixels(M, Ci, Co)
        nparts = max(1, min(1024, (M + tp - 1) // tp))
    if os.environ.get("KB_PARTS"):
        nparts = int(os.environ["KB_PARTS"])
    stats = torch.full((2, Ci, nparts), float("nan"), device="cuda")
    def call(i):
        a = L.MnasConvGemm()
        a.mode, a.N, a.Hi, a.Wi, a.Ci, a.Ho, a.Wo, a.Co = 1, N, H, H, Co, H, H, Ci
        a.kh = a.kw = 1; a.stride, a.pad, a.nparts = 1, 0, nparts
        a.grad = L.MnasGradIn(gs[i].data_ptr(), ys[i].data_ptr(), coef.data_ptr())
        a.w, a.out, a.stats = wp.data_ptr(), outs[i].data_ptr(), stats.data_ptr()
        a.red_y, a.red_bn = rys[i].data_ptr(), rbn.data_ptr()
        L.check(lib.mnas_conv_gemm(C.byref(a), L.cur_stream()), "conv_gemm")
    for i in range(nset): call(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 3 * nset
    e0.record()
    for i in range(iters): call(i % nset)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print("pw dgrad H=%3d conv %4d->%4d nparts=%4d : %7.1f us  %5.2f TB/s" %
          (H, Ci, Co, nparts, us, M * (2 * Co + 2 * Ci) * 2 / us / 1e6), flush=True)
sel = os.environ.get("KB_SHAPES")
for s in SHAPES:
    if sel and ("%d-%d" % (s[1], s[2])) not in sel.split(","):
        continue
    run(*s)
