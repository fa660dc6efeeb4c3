"""Stand-alone chain benchmark of ONE inverted-residual block application (mnasnet.py:105-137) on the bandwidth-bound maps:
the per-layer launches the engine runs today against the spatially tiled fused chain (round 4).  The BatchNorm finalize /
bookkeeping launches (~5 us each, identical in number) are left out of both chains.

    python tools/kbench_irb112.py [H C E k [N]]        default 112 16 48 3 256

per-layer chain   fwd: expand 1x1 (k_pwf) -> depthwise (k_dw_fwd) -> project 1x1 (k_igemm/k_pws) -> add_act
                  bwd: project backward (k_pw_bwd) -> depthwise backward (k_dw_bwd) -> expand backward (k_pw_bwd)
fused chain       fwd: Gram statistics (k_gram) -> expand+depthwise (k_dw_fwd_exp, y1 never stored) -> project -> add_act
                  bwd: project backward without the g2 store (k_pw_bwd NOGIN, dy3 materialised) -> depthwise backward with
                       the g2 / y1 rings recomputed on the matrix cores (k_dw_bwd SRC) -> expand backward with y1 recomputed
                       from x (k_pw_bwd RECOMP)
Every launch is timed alone with HIP events over rotating buffer sets (nothing survives in the 256 MiB Infinity Cache), then
the two chains back to back.  Results are NOT checked here (tests/test_gpu_tiled.py does that)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from mnasnet_pytorch_amd import _lib as L

lib = L.load()
argv = [int(v) for v in sys.argv[1:]]
H, Cc, E, k = (argv + [112, 16, 48, 3][len(argv):])[:4] if len(argv) < 4 else argv[:4]
N = argv[4] if len(argv) > 4 else 256
W = H
M = N * H * W
dev = "cuda"
NSET = 2
S_MB, L_MB = M * Cc * 2 / 1e6, M * E * 2 / 1e6


def bf(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


def coefs(Cn):
    b = torch.zeros(8, Cn, device=dev)
    b[0] = 1.0 + 0.3 * torch.rand(Cn, device=dev)
    b[1] = 0.2 * torch.randn(Cn, device=dev)
    b[2] = b[0]
    b[3] = 0.05 * torch.randn(Cn, device=dev)
    b[4] = 0.02 * torch.randn(Cn, device=dev)
    b[5] = 0.1 * torch.randn(Cn, device=dev)
    b[6] = 1.0 + 0.2 * torch.rand(Cn, device=dev)
    return b


def pack(w, kind):
    Co, Ci, kh, kw = w.shape
    dst = torch.empty(lib.mnas_packed_bytes(kind, Co, Ci, kh, kw), dtype=torch.uint8, device=dev)
    L.check(lib.mnas_pack_weights(w.data_ptr(), kind, Co, Ci, kh, kw, dst.data_ptr(), L.cur_stream()))
    return dst


class Set:
    def __init__(self):
        self.x = bf(N, H, W, Cc)            # block input (materialised sum of the previous block)
        self.y1, self.y2 = bf(N, H, W, E), bf(N, H, W, E)
        self.y3, self.out = bf(N, H, W, Cc), bf(N, H, W, Cc)
        self.G = bf(N, H, W, Cc)            # gradient wrt the block output
        self.g2, self.g1 = bf(N, H, W, E), bf(N, H, W, E)
        self.gx = bf(N, H, W, Cc)
        self.dy3 = bf(N, H, W, Cc)
        self.yprev = bf(N, H, W, Cc)        # raw output of the previous block's project conv (fused-reduce target of gx)


sets = [Set() for _ in range(NSET)]
w1 = torch.randn(E, Cc, 1, 1, device=dev) / Cc ** 0.5
w3 = torch.randn(Cc, E, 1, 1, device=dev) / E ** 0.5
wd = torch.randn(E, 1, k, k, device=dev) / k
b1, bd_, b3 = (0.1 * torch.randn(n, device=dev) for n in (E, E, Cc))
w1f, w1d, w3f, w3d, wdp = pack(w1, L.PACK_FWD), pack(w1, L.PACK_DGRAD), pack(w3, L.PACK_FWD), pack(w3, L.PACK_DGRAD), pack(wd, L.PACK_DW)
bn1, bn2, bn3, bnp = coefs(E), coefs(E), coefs(Cc), coefs(Cc)
stats = torch.empty(2 * 2048 * max(E, 64), device=dev)
red = torch.empty(2 * 2048 * max(E, 64), device=dev)
wsc = [torch.empty(1024 * max(E * Cc, k * k * E), device=dev) for _ in range(3)]
AI = L.MnasActIn
GI = L.MnasGradIn
st = L.cur_stream


def gemm_parts(Mx, Ci, Co):
    p = lib.mnas_conv_gemm_parts(0, Mx, Ci, Co, 1)
    if p < 1:
        p = max(1, min(1024, (Mx + 127) // 128))
    return p


def k_expand(s):
    a = L.MnasConvGemm()
    a.mode, a.N, a.Hi, a.Wi, a.Ci, a.Ho, a.Wo, a.Co = 0, N, H, W, Cc, H, W, E
    a.kh = a.kw = 1; a.stride, a.pad, a.nparts = 1, 0, gemm_parts(M, Cc, E)
    a.act = AI(s.x.data_ptr(), None, None)
    a.w, a.bias, a.out, a.stats = w1f.data_ptr(), b1.data_ptr(), s.y1.data_ptr(), stats.data_ptr()
    L.check(lib.mnas_conv_gemm(C.byref(a), st()), "expand")


def k_project(s):
    a = L.MnasConvGemm()
    a.mode, a.N, a.Hi, a.Wi, a.Ci, a.Ho, a.Wo, a.Co = 0, N, H, W, E, H, W, Cc
    a.kh = a.kw = 1; a.stride, a.pad, a.nparts = 1, 0, gemm_parts(M, E, Cc)
    a.act = AI(s.y2.data_ptr(), bn2[0].data_ptr(), bn2[1].data_ptr())
    a.w, a.bias, a.out, a.stats = w3f.data_ptr(), b3.data_ptr(), s.y3.data_ptr(), stats.data_ptr()
    L.check(lib.mnas_conv_gemm(C.byref(a), st()), "project")


NL = max(64, min(2048, (M * E + 8191) // 8192))


def k_dw_fwd(s):
    a = L.MnasDwFwd()
    a.N, a.H, a.W, a.C, a.k, a.nparts = N, H, W, E, k, NL
    a.in_ = AI(s.y1.data_ptr(), bn1[0].data_ptr(), bn1[1].data_ptr())
    a.w, a.bias, a.out, a.stats = wdp.data_ptr(), bd_.data_ptr(), s.y2.data_ptr(), stats.data_ptr()
    L.check(lib.mnas_dw_fwd(C.byref(a), st()), "dw_fwd")


def k_add(s):
    A, B = AI(s.x.data_ptr(), None, None), AI(s.y3.data_ptr(), bn3[0].data_ptr(), bn3[1].data_ptr())
    L.check(lib.mnas_add_act(C.byref(A), C.byref(B), M, Cc, s.out.data_ptr(), None, H * W, st()), "add_act")


def k_pwb_proj(s, nogin=False):
    c = L.MnasPwBwd()
    c.M, c.Ci, c.Co, c.nparts = M, E, Cc, 1024
    c.x = AI(s.y2.data_ptr(), bn2[0].data_ptr(), bn2[1].data_ptr())
    c.dy = GI(s.G.data_ptr(), s.y3.data_ptr(), bn3.data_ptr())
    c.w, c.gin, c.wpartial = w3d.data_ptr(), s.g2.data_ptr(), wsc[0].data_ptr()
    c.red_partial, c.red_y, c.red_bn = red.data_ptr(), s.y2.data_ptr(), bn2.data_ptr()
    if nogin:
        c.gin = None
        c.dy_out = s.dy3.data_ptr()
    L.check(lib.mnas_pw_bwd(C.byref(c), st()), "pw_bwd proj")


NB = max(64, min(1024, (M * E + 8191) // 8192))


def k_dw_bwd(s):
    a = L.MnasDwBwd()
    a.N, a.H, a.W, a.C, a.k, a.nparts = N, H, W, E, k, NB
    a.x, a.dy = AI(s.y1.data_ptr(), bn1[0].data_ptr(), bn1[1].data_ptr()), GI(s.g2.data_ptr(), s.y2.data_ptr(), bn2.data_ptr())
    a.w, a.gin, a.wpartial = wdp.data_ptr(), s.g1.data_ptr(), wsc[1].data_ptr()
    a.red_bn, a.red_partial, a.phase = bn1.data_ptr(), red.data_ptr(), 0
    L.check(lib.mnas_dw_bwd(C.byref(a), st()), "dw_bwd")


def k_pwb_exp(s, recomp=False):
    c = L.MnasPwBwd()
    c.M, c.Ci, c.Co, c.nparts = M, Cc, E, 1024
    c.x = AI(s.x.data_ptr(), None, None)
    c.dy = GI(s.g1.data_ptr(), None if recomp else s.y1.data_ptr(), bn1.data_ptr())
    c.w, c.gin, c.wpartial, c.resid = w1d.data_ptr(), s.gx.data_ptr(), wsc[2].data_ptr(), s.G.data_ptr()
    c.red_partial, c.red_y, c.red_bn = red.data_ptr(), s.yprev.data_ptr(), bnp.data_ptr()
    if recomp:
        c.w_fwd, c.b_fwd = w1f.data_ptr(), b1.data_ptr()
    L.check(lib.mnas_pw_bwd(C.byref(c), st()), "pw_bwd expand")


gram_nsplit = max(1, min(512, (M + 2047) // 2048))
gp = torch.empty(gram_nsplit * Cc * Cc, device=dev)
sp = torch.empty(gram_nsplit * Cc, device=dev)


def k_gram(s):
    ai = AI(s.x.data_ptr(), None, None)
    L.check(lib.mnas_gram(C.byref(ai), M, Cc, gram_nsplit, gp.data_ptr(), sp.data_ptr(), st()), "gram")


def k_dw_exp(s, keep_y1=False):
    f = L.MnasDwExpFwd()
    f.N, f.H, f.W, f.C, f.k, f.Cin, f.nparts = N, H, W, E, k, Cc, NL
    f.x = AI(s.x.data_ptr(), None, None)
    f.w1, f.b1, f.bn1_scale, f.bn1_shift = w1f.data_ptr(), b1.data_ptr(), bn1[0].data_ptr(), bn1[1].data_ptr()
    f.w, f.bias, f.y1, f.out, f.stats = wdp.data_ptr(), bd_.data_ptr(), (s.y1.data_ptr() if keep_y1 else None), s.y2.data_ptr(), stats.data_ptr()
    L.check(lib.mnas_dw_exp_fwd(C.byref(f), st()), "dw_exp_fwd")


def k_dw_bwd_src(s):
    a = L.MnasDwBwd()
    a.N, a.H, a.W, a.C, a.k, a.nparts = N, H, W, E, k, NB
    a.x, a.dy = AI(None, bn1[0].data_ptr(), bn1[1].data_ptr()), GI(None, s.y2.data_ptr(), bn2.data_ptr())
    a.w, a.gin, a.wpartial = wdp.data_ptr(), s.g1.data_ptr(), wsc[1].data_ptr()
    a.red_bn, a.red_partial, a.phase = bn1.data_ptr(), red.data_ptr(), 0
    a.src_x = AI(s.x.data_ptr(), None, None)
    a.src_w1, a.src_b1, a.src_dy, a.src_w3t, a.src_cin = w1f.data_ptr(), b1.data_ptr(), s.dy3.data_ptr(), w3d.data_ptr(), Cc
    L.check(lib.mnas_dw_bwd(C.byref(a), st()), "dw_bwd src")


def timeit(fn, mb, name, iters=6):
    try:
        for s in sets:
            fn(s)
    except Exception as e:          # kernel form not built (yet) / unsupported shape
        print("%-34s unavailable: %s" % (name, e), flush=True)
        return None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(sets[i % NSET])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print("%-34s %8.1f us   %7.0f MB   %5.2f TB/s" % (name, us, mb, mb / us), flush=True)
    return us


print("block application  N=%d %dx%d  C=%d E=%d k=%d   S=%.0f MB  L=%.0f MB" % (N, H, W, Cc, E, k, S_MB, L_MB))
S_, L_ = S_MB, L_MB
HAS = lambda cls, f: any(n == f for n, _ in cls._fields_)
base = [("expand 1x1", k_expand, S_ + L_), ("depthwise fwd", k_dw_fwd, 2 * L_), ("project 1x1", k_project, L_ + S_), ("add_act", k_add, 3 * S_),
        ("project bwd (k_pw_bwd)", k_pwb_proj, 2 * S_ + 2 * L_), ("depthwise bwd", k_dw_bwd, 4 * L_),
        ("expand bwd (k_pw_bwd)", k_pwb_exp, 2 * L_ + 4 * S_)]
tb = [timeit(f, mb, "base  " + n) for n, f, mb in base]
fused = [("gram", k_gram, S_), ("expand+depthwise (no y1)", k_dw_exp, S_ + L_), ("project 1x1", k_project, L_ + S_), ("add_act", k_add, 3 * S_)]
if HAS(L.MnasPwBwd, "dy_out"):
    fused.append(("project bwd NOGIN", lambda s: k_pwb_proj(s, True), 3 * S_ + L_))
if HAS(L.MnasDwBwd, "src_dy"):
    fused.append(("depthwise bwd SRC", k_dw_bwd_src, 2 * S_ + 2 * L_))
if HAS(L.MnasPwBwd, "w_fwd"):
    fused.append(("expand bwd RECOMP", lambda s: k_pwb_exp(s, True), L_ + 4 * S_))
tf = [timeit(f, mb, "fused " + n) for n, f, mb in fused]
timeit(lambda s: k_dw_exp(s, True), S_ + 2 * L_, "      expand+depthwise (+y1 store)")
if all(t is not None for t in tb):
    print("per-layer chain: %.1f us (sum of launches)" % sum(tb))
if len(fused) == 7 and all(t is not None for t in tf):
    print("fused chain    : %.1f us (sum of launches)   speedup %.3fx" % (sum(tf), sum(tb) / sum(tf)))


    def chain(fs):
        def run(s):
            for f in fs:
                f(s)
        return run
    a = timeit(chain([f for _, f, _ in base]), 0, "per-layer chain back to back")
    b = timeit(chain([f for _, f, _ in fused]), 0, "fused chain back to back")
    if a and b:
        print("back-to-back speedup %.3fx" % (a / b))
