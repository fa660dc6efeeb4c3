"""Micro-benchmark of mnas_conv_wgrad at the 14 launch shapes of the bs-256 bench step (the 14x14 / 7x7 pointwise convs and the
dense 3x3 convs): us per launch ALONE on the device (in the step they run on the side stream next to the main chain) and
algorithmic TFLOP/s.  Buffers rotate so that nothing stays in the Infinity Cache between iterations.  KB_WGS = workgroup budget
(Engine.wgrad_wgs)."""
import ctypes as C, sys, os, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from mnasnet_pytorch_amd import _lib as L
lib = L.load()
N = int(os.environ.get("KB_N", "256"))
WGS = int(os.environ.get("KB_WGS", "512"))
FIN = int(os.environ.get("KB_FIN", "1"))       # time the split reduction (mnas_wgrad_finalize) with the launch
SHAPES = [  # H, Ci, Co, k, stride, dy materialised
    (7, 192, 320, 3, 1, 1), (7, 1152, 192, 1, 1, 0), (7, 192, 1152, 1, 1, 0), (14, 96, 192, 3, 2, 1), (14, 96, 576, 1, 1, 0),
    (14, 80, 96, 3, 1, 1), (14, 80, 480, 1, 1, 0), (28, 40, 80, 3, 2, 1), (56, 24, 40, 3, 2, 1), (112, 16, 24, 3, 2, 1)]


def cdiv(a, b): return (a + b - 1) // b


def run(H, Ci, Co, k, s, mat):
    pad = k // 2
    Ho = (H + 2 * pad - k) // s + 1
    M = N * Ho * Ho
    K = k * k * Ci
    nset = max(2, min(6, int(400e6 / (N * H * H * Ci * 2 + M * Co * 4)) + 1))
    bf = lambda *sh: (torch.randn(*sh, device="cuda") * 0.5).to(torch.bfloat16)
    xs = [bf(N * H * H, Ci) for _ in range(nset)]
    gs = [bf(M, Co) for _ in range(nset)]
    ys = [bf(M, Co) for _ in range(nset)]
    sc, sh = torch.rand(Ci, device="cuda") + 0.5, torch.randn(Ci, device="cuda") * 0.2
    coef = torch.randn(8, Co, device="cuda")
    slabs = lib.mnas_conv_wgrad_slabs(Co, Ci, k * k) if hasattr(lib, "mnas_conv_wgrad_slabs") else cdiv(Co, 64) * cdiv(K, 64)
    nsp = max(1, min(cdiv(WGS, slabs) if os.environ.get('KB_CEIL') else WGS // slabs, cdiv(M, 256)))
    partial = torch.empty(nsp * Co * K, device="cuda")
    def call(i):
        a = L.MnasConvWgrad()
        a.N, a.Hi, a.Wi, a.Ci, a.Ho, a.Wo, a.Co = N, H, H, Ci, Ho, Ho, Co
        a.kh = a.kw = k
        a.stride, a.pad, a.nsplit = s, pad, nsp
        a.x = L.MnasActIn(xs[i].data_ptr(), sc.data_ptr(), sh.data_ptr())
        a.dy = L.MnasGradIn(gs[i].data_ptr(), None, None) if mat else L.MnasGradIn(gs[i].data_ptr(), ys[i].data_ptr(), coef.data_ptr())
        a.partial = partial.data_ptr()
        L.check(lib.mnas_conv_wgrad(C.byref(a), L.cur_stream()), "wgrad")
        if FIN: L.check(lib.mnas_wgrad_finalize(partial.data_ptr(), nsp, Co, Ci, k * k, grad.data_ptr(), 1, L.cur_stream()), "fin")
    grad = torch.zeros(Co * K, device="cuda")
    for i in range(nset): call(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 4 * nset
    e0.record()
    for i in range(iters): call(i % nset)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    fl = 2.0 * M * K * Co
    print("%3dx%-3d %4d->%-4d k%d s%d  slabs %3d nsp %3d  %7.1f us  %6.1f TFLOP/s" % (H, H, Ci, Co, k, s, slabs, nsp, us, fl / us / 1e6))
    return us


tot = sum(run(*s) * (4 if s[:3] == (14, 96, 576) else 2 if s[:3] == (14, 80, 480) else 1) for s in SHAPES)
print("sum over the step's 14 launches: %.1f us" % tot)
