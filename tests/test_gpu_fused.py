"""-m gpu: the fused expand+depthwise path (csrc/mnas_gram.hip, mnas_dw.hip EXP forms) through the C ABI.

* mnas_gram + mnas_gram_bn_finalize: BatchNorm statistics of y = W act(x) + b derived from the second moments of act(x),
  against fp64 statistics of the explicitly computed y (same bf16-rounded operands).  Tolerance 2e-3 on scale / invstd
  (a quadratic form of fp32-accumulated moments; measured ~1e-5), 2e-3 * |ref|max on mean / shift.
* mnas_dw_exp_fwd against the unfused pair mnas_conv_gemm(mode 0) -> mnas_dw_fwd with the SAME BatchNorm coefficients: the
  expanded tensor must be bit-identical (same MFMA k order), the depthwise output bit-identical, the statistics equal to 2e-3
  (different strip geometry = different summation order)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from cases import O
from gpu_util import L, act_in, bf16r, conv_gemm, from_nhwc, nhwc, pack, relerr

pytestmark = pytest.mark.gpu


def _x(shape, seed):
    return bf16r(O.det_uniform(shape, seed))


GRAM = [(2, 12, 12, 16, 48, True), (3, 9, 7, 24, 72, False), (2, 14, 14, 96, 576, True), (1, 7, 7, 192, 1152, True),
        (4, 28, 28, 40, 240, True), (2, 10, 10, 80, 480, False)]


@pytest.mark.parametrize("shape", GRAM)
@pytest.mark.parametrize("nsplit", [1, 7])
def test_gram_bn(shape, nsplit):
    lib = L.load()
    N, H, W, Ci, Co, virt = shape
    M = N * H * W
    x = _x((N, Ci, H, W), 1) + (0.3 if not virt else 0.0)
    w = O.det_param("t.conv.weight", (Co, Ci, 1, 1), 2)
    bias = 0.1 * O.det_uniform((Co,), 3)
    sc, sh = 1 + 0.3 * O.det_uniform((Ci,), 4), 0.2 * O.det_uniform((Ci,), 5)
    a = bf16r(F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))) if virt else bf16r(x)
    y = F.conv2d(a.double(), bf16r(w).double(), bias.double())
    mean = y.mean((0, 2, 3))
    var = y.var((0, 2, 3), unbiased=False)
    gamma, beta = 1 + 0.2 * O.det_uniform((Co,), 6), 0.1 * O.det_uniform((Co,), 7)
    rm, rv = 0.1 * O.det_uniform((Co,), 8), 1 + 0.3 * O.det_uniform((Co,), 9).abs()
    xd = nhwc(x)
    gp = torch.full((nsplit, Ci, Ci), float("nan"), device="cuda")
    sp = torch.full((nsplit, Ci), float("nan"), device="cuda")
    ai = act_in(xd, sc.cuda(), sh.cuda()) if virt else act_in(xd)
    L.check(lib.mnas_gram(C.byref(ai), M, Ci, nsplit, gp.data_ptr(), sp.data_ptr(), L.cur_stream()), "gram")
    am = a.permute(0, 2, 3, 1).reshape(M, Ci).double()
    assert relerr(gp.cpu().double().sum(0), am.t() @ am) < 2e-3
    assert relerr(sp.cpu().double().sum(0), am.sum(0)) < 2e-3
    d = lambda t: t.clone().float().cuda()
    dw, db, dg, dbe, drm, drv = d(w.view(Co, Ci)), d(bias), d(gamma), d(beta), d(rm), d(rv)
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    bn = torch.zeros(8, Co, device="cuda")
    scratch = torch.empty(Ci * Ci + Ci, dtype=torch.float64, device="cuda")
    L.check(lib.mnas_gram_bn_finalize(gp.data_ptr(), sp.data_ptr(), nsplit, Ci, Co, float(M), dw.data_ptr(), db.data_ptr(),
                                      dg.data_ptr(), dbe.data_ptr(), drm.data_ptr(), drv.data_ptr(), nbt.data_ptr(), 0.1, 1e-5,
                                      scratch.data_ptr(), bn.data_ptr(), L.cur_stream()), "gram_bn")
    invstd = 1 / torch.sqrt(var + 1e-5)
    s = gamma.double() * invstd
    t = beta.double() - mean * s
    bn = bn.cpu().double()
    assert relerr(bn[6], invstd) < 2e-3 and relerr(bn[0], s) < 2e-3
    assert relerr(bn[5], mean) < 2e-3 and relerr(bn[1], t) < 2e-3
    assert relerr(drm.cpu(), 0.9 * rm.double() + 0.1 * mean) < 2e-3
    assert relerr(drv.cpu(), 0.9 * rv.double() + 0.1 * var * M / (M - 1)) < 2e-3
    assert int(nbt) == 1


EXP = [  # N,H,W,Cin,C,k
    (2, 12, 12, 16, 48, 3), (2, 20, 20, 24, 72, 5), (2, 28, 28, 40, 240, 5), (3, 14, 14, 80, 480, 3), (2, 14, 14, 96, 576, 5),
    (1, 45, 37, 16, 48, 3), (2, 9, 50, 24, 72, 5), (1, 112, 112, 16, 48, 3),
]


@pytest.mark.parametrize("shape", EXP)
@pytest.mark.parametrize("virt", [True, False])
@pytest.mark.parametrize("keep_y1", [True, False])
def test_dw_exp_fwd(shape, virt, keep_y1):
    lib = L.load()
    N, H, W, Ci, C_, k = shape
    x = _x((N, Ci, H, W), 1)
    w1 = bf16r(O.det_param("t.conv.weight", (C_, Ci, 1, 1), 2))
    b1 = 0.1 * O.det_uniform((C_,), 3)
    xs, xt = 1 + 0.3 * O.det_uniform((Ci,), 4), 0.2 * O.det_uniform((Ci,), 5)
    wd = O.det_param("t.dw.weight", (C_, 1, k, k), 6)
    bd = 0.1 * O.det_uniform((C_,), 7)
    s1, t1 = 1 + 0.3 * O.det_uniform((C_,), 8), 0.2 * O.det_uniform((C_,), 9)      # BatchNorm of the expand conv
    xd = nhwc(x)
    dxs, dxt, ds1, dt1 = xs.cuda(), xt.cuda(), s1.cuda(), t1.cuda()
    ain = act_in(xd, dxs, dxt) if virt else act_in(xd)
    w1p, wdp = pack(w1, L.PACK_FWD), pack(wd, L.PACK_DW)
    db1, dbd = b1.cuda(), bd.cuda()
    # ---- unfused pair
    y1_ref, _ = conv_gemm(0, N, H, W, Ci, H, W, C_, 1, 1, 0, w1p, bias=db1, act=ain, nparts=max(1, min(64, N * H * W // 64)))
    nparts = 40
    out_ref = torch.empty((N, H, W, C_), dtype=torch.bfloat16, device="cuda")
    rows = lib.mnas_dw_rows(N, H, W, C_, k, nparts, 0)
    st_ref = torch.full((2, C_, rows), float("nan"), device="cuda")
    a_ = L.MnasDwFwd()
    a_.N, a_.H, a_.W, a_.C, a_.k, a_.nparts = N, H, W, C_, k, nparts
    a_.in_ = act_in(y1_ref, ds1, dt1)
    a_.w, a_.bias, a_.out, a_.stats = wdp.data_ptr(), dbd.data_ptr(), out_ref.data_ptr(), st_ref.data_ptr()
    L.check(lib.mnas_dw_fwd(C.byref(a_), L.cur_stream()), "dw_fwd")
    # ---- fused
    rows_f = lib.mnas_dw_exp_rows(N, H, W, C_, k, Ci, nparts)
    assert 1 <= rows_f <= nparts
    out = torch.full((N, H, W, C_), float("nan"), dtype=torch.bfloat16, device="cuda")
    y1 = torch.full((N, H, W, C_), float("nan"), dtype=torch.bfloat16, device="cuda") if keep_y1 else None
    st = torch.full((2, C_, rows_f), float("nan"), device="cuda")
    f = L.MnasDwExpFwd()
    f.N, f.H, f.W, f.C, f.k, f.Cin, f.nparts = N, H, W, C_, k, Ci, nparts
    f.x = ain
    f.w1, f.b1, f.bn1_scale, f.bn1_shift = w1p.data_ptr(), db1.data_ptr(), ds1.data_ptr(), dt1.data_ptr()
    f.w, f.bias, f.y1, f.out, f.stats = wdp.data_ptr(), dbd.data_ptr(), L.ptr(y1), out.data_ptr(), st.data_ptr()
    L.check(lib.mnas_dw_exp_fwd(C.byref(f), L.cur_stream()), "dw_exp_fwd")
    torch.cuda.synchronize()
    if keep_y1:
        assert torch.equal(y1.view(torch.int16), y1_ref.view(torch.int16)), "expanded tensor differs from the unfused conv"
    assert torch.equal(out.view(torch.int16), out_ref.view(torch.int16)), \
        "depthwise output differs: max |d| = %g" % float((out.float() - out_ref.float()).abs().max())
    assert relerr(st.cpu().double().sum(-1), st_ref.cpu().double().sum(-1)) < 2e-3


TCONV = [  # N,Ho,Wo,Co,Ci  (forward conv: Ci -> Co, 3x3 stride 2 pad 1, input plane 2Ho x 2Wo)
    (2, 6, 6, 24, 16), (3, 5, 7, 32, 16), (2, 28, 28, 24, 16), (1, 56, 56, 24, 16), (5, 9, 4, 16, 8), (2, 7, 9, 24, 24),
    # the 112x112 / 56x56 stage transitions on the weight-stationary kernel (csrc/mnas_tcx.hip: 1 / 2 channel tiles per class)
    (2, 28, 28, 40, 24), (7, 5, 9, 40, 24), (3, 6, 5, 32, 32), (1, 3, 2, 8, 8),
    # 96 -> 192 onto the 7x7 plane: K-split register-resident weights over an LDS image of dy (k_tcr), N >= 32
    (33, 7, 7, 192, 96), (40, 7, 6, 192, 112), (70, 5, 8, 192, 96),
]


@pytest.mark.parametrize("shape", TCONV)
@pytest.mark.parametrize("red", [True, False])
def test_tconv_dgrad(shape, red):
    """mnas_tconv_dgrad (transposed convolution over a materialised dy) against torch's conv2d input gradient on the same
    bf16-rounded operands (<= 6e-3 of max|ref|) and, for the fused BatchNorm-backward reduce, against fp64 sums (<= 2e-3)."""
    lib = L.load()
    N, Ho, Wo, Co, Ci = shape
    Hi, Wi = 2 * Ho, 2 * Wo
    dy = _x((N, Co, Ho, Wo), 1)
    w = bf16r(O.det_param("t.conv.weight", (Co, Ci, 3, 3), 2))
    ref = torch.nn.grad.conv2d_input((N, Ci, Hi, Wi), w, dy, stride=2, padding=1)
    dyd = nhwc(dy)
    wp = pack(w, L.PACK_TCONV)
    nparts = lib.mnas_tconv_parts(N, Ho, Wo, Co, Ci)
    assert nparts >= 1 and lib.mnas_tconv_supported(Ho, Wo, Co, Ci)
    out = torch.full((N, Hi, Wi, Ci), float("nan"), dtype=torch.bfloat16, device="cuda")
    a = L.MnasTconvDgrad()
    a.N, a.Ho, a.Wo, a.Co, a.Ci, a.nparts = N, Ho, Wo, Co, Ci, nparts
    a.dy, a.w, a.out = dyd.data_ptr(), wp.data_ptr(), out.data_ptr()
    if red:
        from gpu_util import rand_bn_coefs
        y_t = _x((N, Ci, Hi, Wi), 5)
        b = rand_bn_coefs(Ci, 9, O)
        ytd, bd = nhwc(y_t), b.cuda()
        st = torch.full((2, Ci, nparts), float("nan"), device="cuda")
        a.stats, a.red_y, a.red_bn = st.data_ptr(), ytd.data_ptr(), bd.data_ptr()
    L.check(lib.mnas_tconv_dgrad(C.byref(a), L.cur_stream()), "tconv_dgrad")
    got = from_nhwc(out)
    assert relerr(got, ref) < 6e-3
    if red:
        g = bf16r(got)                                   # the reduce uses the gradient as stored
        s_, t_ = b[0].view(1, -1, 1, 1), b[1].view(1, -1, 1, 1)
        dz = (g * ((s_ * y_t + t_) > 0)).double()
        xhat = ((y_t - b[5].view(1, -1, 1, 1)) * b[6].view(1, -1, 1, 1)).double()
        tot = st.cpu().double().sum(-1)
        assert relerr(tot[0], dz.sum((0, 2, 3))) < 2e-3
        assert relerr(tot[1], (dz * xhat).sum((0, 2, 3))) < 2e-3
