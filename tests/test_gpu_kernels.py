"""-m gpu: every HIP kernel, called through the C ABI, against fp32 torch-CPU math on the same bf16-rounded
inputs.  Tolerance: outputs are stored as bf16 (half-ulp 2^-9) after fp32 accumulation, so the bound is
max|hip - ref| <= 6e-3 * max|ref| for bf16 tensors and 2e-3 for fp32 reductions (statistics, weight
gradients; limited by bf16 rounding of the staged operands, which the reference reproduces)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from cases import O
from gpu_util import (L, act_in, bf16r, conv_gemm, dy_ref, from_nhwc, grad_in, nhwc, pack, rand_bn_coefs, relerr)

pytestmark = pytest.mark.gpu
TOL_BF16 = 6e-3
TOL_F32 = 2e-3


def _x(shape, seed):
    return bf16r(O.det_uniform(shape, seed))


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C_,nparts,count", [(16, 7, 100.0), (48, 64, 3211264.0), (1152, 33, 12544.0),
                                             (24, 1031, 802816.0), (48, 2048, 3211264.0), (1152, 300, 12544.0)])   # block-per-channel path
def test_bn_fwd_finalize(C_, nparts, count):
    lib = L.load()
    u = O.det_uniform((2, C_, nparts), 3)
    per = count / nparts
    partial = torch.empty(2, C_, nparts)          # channel-major partial layout
    mean_p = 0.3 * u[0]
    partial[0] = per * mean_p
    partial[1] = per * (mean_p ** 2 + 0.5 + 0.4 * u[1].abs())
    gamma, beta = 1 + 0.2 * O.det_uniform((C_,), 4), 0.1 * O.det_uniform((C_,), 5)
    rm, rv = 0.1 * O.det_uniform((C_,), 6), 1 + 0.3 * O.det_uniform((C_,), 7).abs()
    d = lambda t: t.clone().cuda()
    dp, dg, db, drm, drv = d(partial), d(gamma), d(beta), d(rm), d(rv)
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    bn = torch.zeros(8, C_, device="cuda")
    L.check(lib.mnas_bn_fwd_finalize(dp.data_ptr(), nparts, C_, count, dg.data_ptr(), db.data_ptr(), drm.data_ptr(),
                                     drv.data_ptr(), nbt.data_ptr(), 0.1, 1e-5, 1, bn.data_ptr(), L.cur_stream()))
    S1, S2 = partial[0].double().sum(-1), partial[1].double().sum(-1)
    mean = S1 / count
    var = S2 / count - mean * mean
    invstd = 1 / torch.sqrt(var + 1e-5)
    s = gamma.double() * invstd
    t = beta.double() - mean * s
    bn = bn.cpu().double()
    assert relerr(bn[0], s) < 1e-5 and relerr(bn[1], t) < 1e-5
    assert relerr(bn[5], mean) < 1e-5 and relerr(bn[6], invstd) < 1e-5
    assert relerr(drm.cpu(), 0.9 * rm.double() + 0.1 * mean) < 1e-5
    assert relerr(drv.cpu(), 0.9 * rv.double() + 0.1 * var * count / (count - 1)) < 1e-5
    assert int(nbt) == 1
    # eval mode: coefficients from the running stats, nothing updated
    bn2 = torch.zeros(8, C_, device="cuda")
    L.check(lib.mnas_bn_fwd_finalize(0, 0, C_, count, dg.data_ptr(), db.data_ptr(), drm.data_ptr(), drv.data_ptr(),
                                     0, 0.1, 1e-5, 0, bn2.data_ptr(), L.cur_stream()))
    s_e = gamma / torch.sqrt(drv.cpu() + 1e-5)
    assert relerr(bn2[0].cpu(), s_e) < 1e-5 and relerr(bn2[1].cpu(), beta - drm.cpu() * s_e) < 1e-5


# ---------------------------------------------------------------------------------------------------
PW = [  # N,H,W,Ci,Co
    (2, 12, 12, 16, 48), (2, 12, 12, 48, 16), (3, 9, 7, 72, 24), (2, 6, 7, 96, 576), (2, 5, 5, 1152, 192),
    (5, 16, 16, 32, 16), (2, 7, 7, 240, 40), (1, 28, 28, 40, 240), (2, 14, 14, 576, 96), (1, 9, 9, 480, 80), (3, 5, 5, 120, 40),
    # the widening convs of the <= 28x28 maps: weight-stationary kernel (csrc/mnas_pwx.hip); ragged last pixel group, several
    # groups per workgroup (nparts 13), cout counts that do and do not fill the waves' tiles
    (3, 14, 14, 96, 576), (5, 13, 11, 80, 480), (2, 28, 27, 40, 240), (7, 9, 9, 96, 568), (1, 5, 3, 40, 232), (9, 7, 7, 192, 1152),
]


@pytest.mark.parametrize("shape", PW)
@pytest.mark.parametrize("virt", [True, False])
def test_pw_fwd(shape, virt):
    N, H, W, Ci, Co = shape
    x = _x((N, Ci, H, W), 1)
    w = bf16r(O.det_param("t.conv.weight", (Co, Ci, 1, 1), 2))
    bias = 0.1 * O.det_uniform((Co,), 3)
    sc, sh = 1 + 0.3 * O.det_uniform((Ci,), 4), 0.2 * O.det_uniform((Ci,), 5)
    a = bf16r(F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))) if virt else x
    ref = F.conv2d(a, w, bias)
    dsc, dsh = sc.cuda(), sh.cuda()
    xd = nhwc(x)
    out, st = conv_gemm(0, N, H, W, Ci, H, W, Co, 1, 1, 0, pack(w, L.PACK_FWD), bias.cuda(),
                        act=act_in(xd, dsc if virt else None, dsh if virt else None), nparts=13, stats=True)
    assert relerr(from_nhwc(out), ref) < TOL_BF16
    st = st.cpu().double().sum(-1)
    assert relerr(st[0], ref.double().sum((0, 2, 3))) < TOL_F32
    assert relerr(st[1], (ref.double() ** 2).sum((0, 2, 3))) < TOL_F32


DENSE = [  # N,H,W,Ci,Co,stride
    (2, 12, 12, 16, 24, 2), (2, 7, 9, 80, 96, 1), (2, 9, 9, 24, 40, 2), (1, 7, 7, 192, 320, 1), (2, 14, 10, 96, 192, 2),
    # N >= 32 with an output plane of <= 256 pixels: the whole-image kernel (csrc/mnas_dimg.hip); ragged planes, cout groups
    (32, 14, 14, 80, 96, 1), (33, 14, 14, 96, 192, 2), (32, 7, 7, 192, 320, 1), (40, 9, 11, 24, 40, 2), (32, 16, 16, 16, 24, 1),
    (35, 13, 15, 40, 80, 2),
    # weight-heavy layer on the 7x7 plane: weight slices register-resident, persistent over images (csrc/mnas_c3r.hip)
    (67, 7, 7, 192, 320, 1), (33, 7, 6, 192, 328, 1), (41, 14, 14, 96, 192, 2), (32, 13, 11, 104, 200, 2),
    # stride 2 on the large maps (>= 100 k output pixels): every wave weight-stationary, gather from global (csrc/mnas_c3x.hip)
    (36, 112, 112, 16, 24, 2), (35, 111, 113, 24, 40, 2), (140, 56, 56, 24, 40, 2),
]


@pytest.mark.parametrize("shape", DENSE)
def test_dense_fwd(shape):
    N, H, W, Ci, Co, s = shape
    Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    x = _x((N, Ci, H, W), 1)
    w = bf16r(O.det_param("t.conv.weight", (Co, Ci, 3, 3), 2))
    bias = 0.1 * O.det_uniform((Co,), 3)
    sc, sh = 1 + 0.3 * O.det_uniform((Ci,), 4), 0.2 * O.det_uniform((Ci,), 5)
    a = bf16r(F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)))
    ref = F.conv2d(a, w, bias, stride=s, padding=1)
    dsc, dsh = sc.cuda(), sh.cuda()
    xd = nhwc(x)
    out, st = conv_gemm(0, N, H, W, Ci, Ho, Wo, Co, 3, s, 1, pack(w, L.PACK_FWD), bias.cuda(),
                        act=act_in(xd, dsc, dsh), nparts=5, stats=True)
    assert relerr(from_nhwc(out), ref) < TOL_BF16
    st = st.cpu().double().sum(-1)
    assert relerr(st[0], ref.double().sum((0, 2, 3))) < TOL_F32
    assert relerr(st[1], (ref.double() ** 2).sum((0, 2, 3))) < TOL_F32


@pytest.mark.parametrize("shape", PW)
@pytest.mark.parametrize("with_resid", [True, False])
def test_pw_dgrad(shape, with_resid):
    N, H, W, Ci, Co = shape
    g, y = _x((N, Co, H, W), 1), _x((N, Co, H, W), 2)
    b = rand_bn_coefs(Co, 9, O)
    dy = dy_ref(g, y, b)
    w = bf16r(O.det_param("t.conv.weight", (Co, Ci, 1, 1), 2))
    ref = F.conv_transpose2d(dy, w)
    resid = _x((N, Ci, H, W), 5)
    if with_resid:
        ref = ref + resid
    gd, yd, bd, rd = nhwc(g), nhwc(y), b.cuda(), nhwc(resid)
    # fused BN-backward reduce for the producer of the conv's input: (sum dz, sum dz*xhat) of (out, y_in, bn_in)
    y_in = _x((N, Ci, H, W), 21)
    b_in = rand_bn_coefs(Ci, 22, O)
    yid, bid = nhwc(y_in), b_in.cuda()
    out, st = conv_gemm(1, N, H, W, Co, H, W, Ci, 1, 1, 0, pack(w, L.PACK_DGRAD), None, grad=grad_in(gd, yd, bd),
                        resid=rd if with_resid else None, nparts=11, stats=True, red_y=yid, red_bn=bid)
    assert relerr(from_nhwc(out), ref) < TOL_BF16
    gq = from_nhwc(out)                                  # the reduce sees g as stored (bf16)
    s_, t_, mu_, is_ = (b_in[i].view(1, -1, 1, 1) for i in (0, 1, 5, 6))
    dz = (gq * ((s_ * y_in + t_) > 0)).double()
    xhat = (y_in * is_ - mu_ * is_).double()
    st = st.cpu().double().sum(-1)
    assert relerr(st[0], dz.sum((0, 2, 3))) < 1e-3
    assert relerr(st[1], (dz * xhat).sum((0, 2, 3))) < 1e-3


@pytest.mark.parametrize("shape", [(2, 14, 14, 576, 96), (1, 20, 20, 72, 24), (4, 28, 28, 240, 40)])
@pytest.mark.parametrize("nparts", [1, 8, 40])
def test_pw_dgrad_plain_dy(shape, nparts):
    """materialised dy (no dy-on-load operand) through the DMA-pipelined 1x1 input gradient, several grid sizes"""
    N, H, W, Ci, Co = shape
    dy = _x((N, Co, H, W), 3)
    w = bf16r(O.det_param("t.conv.weight", (Co, Ci, 1, 1), 2))
    ref = F.conv_transpose2d(dy, w)
    g = L.MnasGradIn()
    dyd = nhwc(dy)
    g.g = dyd.data_ptr()
    out, _ = conv_gemm(1, N, H, W, Co, H, W, Ci, 1, 1, 0, pack(w, L.PACK_DGRAD), None, grad=g, nparts=nparts)
    assert relerr(from_nhwc(out), ref) < TOL_BF16


@pytest.mark.parametrize("shape", DENSE)
def test_dense_dgrad(shape):
    N, H, W, Ci, Co, s = shape
    Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    g, y = _x((N, Co, Ho, Wo), 1), _x((N, Co, Ho, Wo), 2)
    b = rand_bn_coefs(Co, 9, O)
    dy = dy_ref(g, y, b)
    w = bf16r(O.det_param("t.conv.weight", (Co, Ci, 3, 3), 2))
    ref = torch.nn.grad.conv2d_input((N, Ci, H, W), w, dy, stride=s, padding=1)
    gd, yd, bd = nhwc(g), nhwc(y), b.cuda()
    out, _ = conv_gemm(1, N, Ho, Wo, Co, H, W, Ci, 3, s, 1, pack(w, L.PACK_DGRAD), None, grad=grad_in(gd, yd, bd), nparts=7)
    assert relerr(from_nhwc(out), ref) < TOL_BF16


@pytest.mark.parametrize("shape", [(32, 14, 14, 80, 96), (32, 7, 7, 192, 320), (36, 9, 11, 24, 40), (32, 16, 16, 16, 24), (2, 14, 14, 80, 96),
                                   (45, 7, 7, 192, 320), (33, 6, 7, 200, 328)])
@pytest.mark.parametrize("nparts", [5, 32])
def test_dense_dgrad_plain_dy_with_reduce(shape, nparts):
    """stride-1 3x3 input gradient over a MATERIALISED dy (what the engine hands the dense convs), fused BatchNorm-backward
    reduce: whole-image kernel for N >= 32, k_igemm otherwise -- same contract"""
    N, H, W, Ci, Co = shape
    dy = _x((N, Co, H, W), 3)
    w = bf16r(O.det_param("t.conv.weight", (Co, Ci, 3, 3), 2))
    ref = torch.nn.grad.conv2d_input((N, Ci, H, W), w, dy, stride=1, padding=1)
    g = L.MnasGradIn()
    dyd = nhwc(dy)
    g.g = dyd.data_ptr()
    y_in = _x((N, Ci, H, W), 21)
    b_in = rand_bn_coefs(Ci, 22, O)
    yid, bid = nhwc(y_in), b_in.cuda()
    out, st = conv_gemm(1, N, H, W, Co, H, W, Ci, 3, 1, 1, pack(w, L.PACK_DGRAD), None, grad=g, nparts=nparts, stats=True,
                        red_y=yid, red_bn=bid)
    assert relerr(from_nhwc(out), ref) < TOL_BF16
    gq = from_nhwc(out)
    s_, t_, mu_, is_ = (b_in[i].view(1, -1, 1, 1) for i in (0, 1, 5, 6))
    dz = (gq * ((s_ * y_in + t_) > 0)).double()
    xhat = (y_in * is_ - mu_ * is_).double()
    st = st.cpu().double().sum(-1)
    assert relerr(st[0], dz.sum((0, 2, 3))) < 1e-3
    assert relerr(st[1], (dz * xhat).sum((0, 2, 3))) < 1e-3


def _wgrad(N, H, W, Ci, Ho, Wo, Co, k, s, pad, xact, dy, nsplit, accumulate=False, init=None):
    lib = L.load()
    K = k * k * Ci
    partial = torch.full((nsplit, Co, K), float("nan"), device="cuda")
    a = L.MnasConvWgrad()
    a.N, a.Hi, a.Wi, a.Ci, a.Ho, a.Wo, a.Co = N, H, W, Ci, Ho, Wo, Co
    a.kh = a.kw = k
    a.stride, a.pad, a.nsplit = s, pad, nsplit
    a.x, a.dy, a.partial = xact, dy, partial.data_ptr()
    L.check(lib.mnas_conv_wgrad(C.byref(a), L.cur_stream()), "wgrad")
    grad = init.clone().cuda() if init is not None else torch.full((Co, Ci, k, k), float("nan"), device="cuda")
    L.check(lib.mnas_wgrad_finalize(partial.data_ptr(), nsplit, Co, Ci, k * k, grad.data_ptr(), int(accumulate),
                                    L.cur_stream()))
    return grad.cpu()


@pytest.mark.parametrize("shape", PW)
def test_pw_wgrad(shape):
    N, H, W, Ci, Co = shape
    x = _x((N, Ci, H, W), 1)
    sc, sh = 1 + 0.3 * O.det_uniform((Ci,), 4), 0.2 * O.det_uniform((Ci,), 5)
    a = bf16r(F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)))
    g, y = _x((N, Co, H, W), 6), _x((N, Co, H, W), 7)
    b = rand_bn_coefs(Co, 9, O)
    dy = dy_ref(g, y, b)
    ref = torch.nn.grad.conv2d_weight(a, (Co, Ci, 1, 1), dy)
    xd, gd, yd, bd, dsc, dsh = nhwc(x), nhwc(g), nhwc(y), b.cuda(), sc.cuda(), sh.cuda()
    got = _wgrad(N, H, W, Ci, H, W, Co, 1, 1, 0, act_in(xd, dsc, dsh), grad_in(gd, yd, bd), 3)
    assert relerr(got, ref) < TOL_F32
    init = O.det_uniform((Co, Ci, 1, 1), 12)
    got2 = _wgrad(N, H, W, Ci, H, W, Co, 1, 1, 0, act_in(xd, dsc, dsh), grad_in(gd, yd, bd), 2, True, init)
    assert relerr(got2, ref + init) < TOL_F32


@pytest.mark.parametrize("shape", DENSE)
def test_dense_wgrad(shape):
    N, H, W, Ci, Co, s = shape
    Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    x = _x((N, Ci, H, W), 1)
    g, y = _x((N, Co, Ho, Wo), 6), _x((N, Co, Ho, Wo), 7)
    b = rand_bn_coefs(Co, 9, O)
    dy = dy_ref(g, y, b)
    ref = torch.nn.grad.conv2d_weight(x, (Co, Ci, 3, 3), dy, stride=s, padding=1)
    xd, gd, yd, bd = nhwc(x), nhwc(g), nhwc(y), b.cuda()
    got = _wgrad(N, H, W, Ci, Ho, Wo, Co, 3, s, 1, act_in(xd), grad_in(gd, yd, bd), 2)
    assert relerr(got, ref) < TOL_F32


@pytest.mark.parametrize("shape,nsplit", [((1, 5, 5, 24, 40, 1), 7), ((2, 7, 7, 192, 320, 1), 3), ((1, 6, 6, 16, 24, 2), 5),
                                          ((3, 7, 7, 8, 168, 1), 2)])
def test_dense_wgrad_slabs_and_empty_splits(shape, nsplit):
    """k_wgrad_t: more pixel splits than 64-pixel chunks (empty splits must contribute zeros), slab shapes with padded edge
    tiles in both dimensions, and mnas_conv_wgrad_slabs consistent with a full cover of dW."""
    N, H, W, Ci, Co, s = shape
    lib = L.load()
    slabs = lib.mnas_conv_wgrad_slabs(Co, Ci, 9)
    assert 1 <= slabs <= ((Co + 31) // 32) * ((9 * Ci + 31) // 32)
    Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    x = _x((N, Ci, H, W), 31)
    g, y = _x((N, Co, Ho, Wo), 36), _x((N, Co, Ho, Wo), 37)
    b = rand_bn_coefs(Co, 39, O)
    dy = dy_ref(g, y, b)
    ref = torch.nn.grad.conv2d_weight(x, (Co, Ci, 3, 3), dy, stride=s, padding=1)
    xd, gd, yd, bd = nhwc(x), nhwc(g), nhwc(y), b.cuda()
    got = _wgrad(N, H, W, Ci, Ho, Wo, Co, 3, s, 1, act_in(xd), grad_in(gd, yd, bd), nsplit)
    assert relerr(got, ref) < TOL_F32
    again = _wgrad(N, H, W, Ci, Ho, Wo, Co, 3, s, 1, act_in(xd), grad_in(gd, yd, bd), nsplit)
    assert torch.equal(got, again)                 # fixed summation order: bit-reproducible


# ---------------------------------------------------------------------------------------------------
DW = [  # N,H,W,C,k
    (2, 12, 12, 48, 3), (2, 12, 12, 72, 5), (2, 7, 9, 240, 5), (3, 14, 14, 480, 3), (5, 7, 7, 1152, 3),
    (2, 33, 20, 32, 3), (2, 28, 28, 72, 5), (3, 7, 7, 576, 5), (1, 40, 24, 120, 5),
    (3, 14, 14, 576, 5), (2, 14, 13, 96, 5),          # the 14-wide planes (forward: 2-column strips, round 6)
]


@pytest.mark.parametrize("shape", DW)
def test_dw_fwd(shape):
    lib = L.load()
    N, H, W, C_, k = shape
    x = _x((N, C_, H, W), 1)
    w = O.det_param("t.conv.weight", (C_, 1, k, k), 2)
    bias = 0.1 * O.det_uniform((C_,), 3)
    sc, sh = 1 + 0.3 * O.det_uniform((C_,), 4), 0.2 * O.det_uniform((C_,), 5)
    a = F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))      # act-on-read: fp32, not re-rounded to bf16
    ref = F.conv2d(a, w, bias, padding=k // 2, groups=C_)
    xd, dsc, dsh, db = nhwc(x), sc.cuda(), sh.cuda(), bias.cuda()
    wp = pack(w, L.PACK_DW)
    nparts = 40         # >= number of channel blocks (C/64)
    out = torch.empty((N, H, W, C_), dtype=torch.bfloat16, device="cuda")
    rows = lib.mnas_dw_rows(N, H, W, C_, k, nparts, 0)
    assert 1 <= rows <= nparts
    st = torch.full((2, C_, rows), float("nan"), device="cuda")
    a_ = L.MnasDwFwd()
    a_.N, a_.H, a_.W, a_.C, a_.k, a_.nparts = N, H, W, C_, k, nparts
    a_.in_ = act_in(xd, dsc, dsh)
    a_.w, a_.bias, a_.out, a_.stats = wp.data_ptr(), db.data_ptr(), out.data_ptr(), st.data_ptr()
    L.check(lib.mnas_dw_fwd(C.byref(a_), L.cur_stream()), "dw_fwd")
    assert relerr(from_nhwc(out), ref) < TOL_BF16
    st = st.cpu().double().sum(-1)
    assert relerr(st[0], ref.double().sum((0, 2, 3))) < TOL_F32
    assert relerr(st[1], (ref.double() ** 2).sum((0, 2, 3))) < TOL_F32


@pytest.mark.parametrize("shape", DW)
@pytest.mark.parametrize("phase", [0, 12])
def test_dw_bwd(shape, phase):
    """phase 0: fused one-sweep backward; 12: the two-launch form (input gradient, then weight gradient)"""
    lib = L.load()
    N, H, W, C_, k = shape
    x = _x((N, C_, H, W), 1)
    w = O.det_param("t.conv.weight", (C_, 1, k, k), 2)
    sc, sh = 1 + 0.3 * O.det_uniform((C_,), 4), 0.2 * O.det_uniform((C_,), 5)
    a = F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))      # act-on-read / dy-on-read: fp32
    g, y = _x((N, C_, H, W), 6), _x((N, C_, H, W), 7)
    b = rand_bn_coefs(C_, 9, O)
    dy = dy_ref(g, y, b, rounded=False)
    ref_gin = torch.nn.grad.conv2d_input((N, C_, H, W), w, dy, padding=k // 2, groups=C_)
    ref_dw = torch.nn.grad.conv2d_weight(a, (C_, 1, k, k), dy, padding=k // 2, groups=C_)
    xd, gd, yd, bd, dsc, dsh = nhwc(x), nhwc(g), nhwc(y), b.cuda(), sc.cuda(), sh.cuda()
    wp = pack(w, L.PACK_DW)
    nparts = 37
    gin = torch.empty((N, H, W, C_), dtype=torch.bfloat16, device="cuda")
    rows1 = lib.mnas_dw_rows(N, H, W, C_, k, nparts, 1 if phase == 0 else 3)     # wpartial: fused or weight-gradient-only launch
    # reduce table: written by the fused launch (3-ring geometry) or by the input-gradient-only launch (2 rings)
    rows0 = rows1 if phase == 0 else lib.mnas_dw_rows(N, H, W, C_, k, nparts, 2)
    wpart = torch.full((rows1, k * k, C_), float("nan"), device="cuda")
    a_ = L.MnasDwBwd()
    a_.N, a_.H, a_.W, a_.C, a_.k, a_.nparts = N, H, W, C_, k, nparts
    a_.x, a_.dy = act_in(xd, dsc, dsh), grad_in(gd, yd, bd)
    a_.w, a_.gin, a_.wpartial = wp.data_ptr(), gin.data_ptr(), wpart.data_ptr()
    # fused BN-backward reduce for the producer of x (x is its raw output, b_in its bnbuf)
    b_in = rand_bn_coefs(C_, 22, O)
    b_in[0], b_in[1] = sc, sh                       # rows 0,1 are the same scale/shift the act-on-load uses
    bid = b_in.cuda()
    redp = torch.full((2, C_, rows0), float("nan"), device="cuda")
    a_.red_bn, a_.red_partial = bid.data_ptr(), redp.data_ptr()
    if phase == 0:
        a_.phase = 0
        L.check(lib.mnas_dw_bwd(C.byref(a_), L.cur_stream()), "dw_bwd")
    else:
        for ph in (1, 2):
            a_.phase = ph
            L.check(lib.mnas_dw_bwd(C.byref(a_), L.cur_stream()), "dw_bwd")
    assert relerr(from_nhwc(gin), ref_gin) < TOL_BF16
    gq = from_nhwc(gin)
    s_, t_, mu_, is_ = (b_in[i].view(1, -1, 1, 1) for i in (0, 1, 5, 6))
    dz = (gq * ((s_ * x + t_) > 0)).double()
    xhat = (x * is_ - mu_ * is_).double()
    rp = redp.cpu().double().sum(-1)
    assert relerr(rp[0], dz.sum((0, 2, 3))) < 1e-3
    assert relerr(rp[1], (dz * xhat).sum((0, 2, 3))) < 1e-3
    grad = torch.full((C_, 1, k, k), float("nan"), device="cuda")
    L.check(lib.mnas_dw_wgrad_finalize(wpart.data_ptr(), rows1, C_, k, grad.data_ptr(), 0, L.cur_stream()))
    assert relerr(grad.cpu(), ref_dw) < TOL_F32


# ---------------------------------------------------------------------------------------------------
# stride-2 depthwise (SepConv(reduce=True), mnasnet.py:73-81; csrc/mnas_dw2.hip): forward + statistics, input gradient, weight
# gradient through mnas_dw_fwd / mnas_dw_bwd with stride = 2; odd planes, 5x5, more than 256 channel pairs (two channel blocks)
@pytest.mark.parametrize("shape", [(2, 12, 12, 48, 3), (3, 13, 11, 72, 5), (2, 9, 9, 600, 3), (5, 7, 8, 32, 5)])
def test_dw_stride2(shape):
    lib = L.load()
    N, H, W, C_, k = shape
    p = k // 2
    Ho, Wo = (H + 2 * p - k) // 2 + 1, (W + 2 * p - k) // 2 + 1
    x = _x((N, C_, H, W), 1)
    w = O.det_param("t.conv.weight", (C_, 1, k, k), 2)
    bias = 0.1 * O.det_uniform((C_,), 3)
    sc, sh = 1 + 0.3 * O.det_uniform((C_,), 4), 0.2 * O.det_uniform((C_,), 5)
    a = F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    ref = F.conv2d(a, w, bias, stride=2, padding=p, groups=C_)
    assert tuple(ref.shape) == (N, C_, Ho, Wo)
    xd, dsc, dsh, db = nhwc(x), sc.cuda(), sh.cuda(), bias.cuda()
    wp = pack(w, L.PACK_DW)
    nparts = 23
    rows = lib.mnas_dw_rows(N, H, W, C_, k, nparts, 4)
    assert 1 <= rows <= nparts
    out = torch.empty((N, Ho, Wo, C_), dtype=torch.bfloat16, device="cuda")
    st = torch.full((2, C_, rows), float("nan"), device="cuda")
    f = L.MnasDwFwd()
    f.N, f.H, f.W, f.C, f.k, f.nparts, f.stride = N, H, W, C_, k, nparts, 2
    f.in_ = act_in(xd, dsc, dsh)
    f.w, f.bias, f.out, f.stats = wp.data_ptr(), db.data_ptr(), out.data_ptr(), st.data_ptr()
    L.check(lib.mnas_dw_fwd(C.byref(f), L.cur_stream()), "dw_fwd stride 2")
    assert relerr(from_nhwc(out), ref) < TOL_BF16
    s1, s2 = st[0].cpu().double().sum(-1), st[1].cpu().double().sum(-1)
    assert relerr(s1, ref.double().sum((0, 2, 3))) < 1e-3 and relerr(s2, (ref.double() ** 2).sum((0, 2, 3))) < 1e-3
    # backward: dy-on-read from (g, y, coef) at the OUTPUT resolution
    g, y = _x((N, C_, Ho, Wo), 6), _x((N, C_, Ho, Wo), 7)
    b = rand_bn_coefs(C_, 9, O)
    dy = dy_ref(g, y, b, rounded=False)
    ref_gin = torch.nn.grad.conv2d_input((N, C_, H, W), w, dy, stride=2, padding=p, groups=C_)
    ref_dw = torch.nn.grad.conv2d_weight(a, (C_, 1, k, k), dy, stride=2, padding=p, groups=C_)
    gd, yd, bd = nhwc(g), nhwc(y), b.cuda()
    gin = torch.empty((N, H, W, C_), dtype=torch.bfloat16, device="cuda")
    wrows = lib.mnas_dw_rows(N, H, W, C_, k, nparts, 7)
    wpart = torch.full((wrows, k * k, C_), float("nan"), device="cuda")
    d = L.MnasDwBwd()
    d.N, d.H, d.W, d.C, d.k, d.nparts, d.stride = N, H, W, C_, k, nparts, 2
    d.x, d.dy = act_in(xd, dsc, dsh), grad_in(gd, yd, bd)
    d.w, d.gin, d.wpartial = wp.data_ptr(), gin.data_ptr(), wpart.data_ptr()
    for ph in (1, 2):
        d.phase = ph
        L.check(lib.mnas_dw_bwd(C.byref(d), L.cur_stream()), "dw_bwd stride 2")
    assert relerr(from_nhwc(gin), ref_gin) < TOL_BF16
    grad = torch.full((C_, 1, k, k), float("nan"), device="cuda")
    L.check(lib.mnas_dw_wgrad_finalize(wpart.data_ptr(), wrows, C_, k, grad.data_ptr(), 0, L.cur_stream()))
    assert relerr(grad.cpu(), ref_dw) < TOL_F32
    d.phase = 0                                        # the fused one-sweep form exists for stride 1 only
    assert lib.mnas_dw_bwd(C.byref(d), L.cur_stream()) == L.EINVAL


# ---------------------------------------------------------------------------------------------------
# (N, H, W[, Co]): W % 4 == 0 with 32 couts runs the band kernels (csrc/mnas_stem.hip; weight gradient also needs Wo % 8 == 0),
# everything else the im2col staging of k_igemm / k_wgrad; partial last bands, odd heights, more bands than workgroups
@pytest.mark.parametrize("shape", [(2, 12, 12), (3, 33, 21), (1, 64, 64), (3, 40, 48), (2, 37, 32), (11, 18, 16), (2, 224, 224),
                                   (2, 24, 24, 16), (1, 34, 80)])
def test_stem(shape):
    lib = L.load()
    N, H, W = shape[:3]
    Co = shape[3] if len(shape) > 3 else 32
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = O.det_uniform((N, 3, H, W), 1)
    w = bf16r(O.det_param("t.conv.weight", (Co, 3, 3, 3), 2))
    bias = 0.1 * O.det_uniform((Co,), 3)
    ref = F.conv2d(bf16r(x), w, bias, stride=2, padding=1)
    xd = x.cuda()
    wp = pack(w.view(Co, 27, 1, 1), L.PACK_FWD)
    nparts = 9
    out = torch.empty((N, Ho, Wo, Co), dtype=torch.bfloat16, device="cuda")
    st = torch.full((2, Co, nparts), float("nan"), device="cuda")
    a = L.MnasStemFwd()
    a.N, a.H, a.W, a.Ho, a.Wo, a.Co, a.nparts = N, H, W, Ho, Wo, Co, nparts
    db = bias.cuda()
    a.x, a.w, a.bias, a.out, a.stats = xd.data_ptr(), wp.data_ptr(), db.data_ptr(), out.data_ptr(), st.data_ptr()
    L.check(lib.mnas_stem_fwd(C.byref(a), L.cur_stream()), "stem_fwd")
    assert relerr(from_nhwc(out), ref) < TOL_BF16
    st = st.cpu().double().sum(-1)
    assert relerr(st[0], ref.double().sum((0, 2, 3))) < TOL_F32
    # wgrad
    g, y = _x((N, Co, Ho, Wo), 6), _x((N, Co, Ho, Wo), 7)
    b = rand_bn_coefs(Co, 9, O)
    dy = dy_ref(g, y, b)
    ref_dw = torch.nn.grad.conv2d_weight(bf16r(x), (Co, 3, 3, 3), dy, stride=2, padding=1)
    gd, yd, bd = nhwc(g), nhwc(y), b.cuda()
    partial = torch.full((nparts, Co, 27), float("nan"), device="cuda")
    s = L.MnasStemWgrad()
    s.N, s.H, s.W, s.Ho, s.Wo, s.Co, s.nparts = N, H, W, Ho, Wo, Co, nparts
    s.x, s.dy, s.partial = xd.data_ptr(), grad_in(gd, yd, bd), partial.data_ptr()
    L.check(lib.mnas_stem_wgrad(C.byref(s), L.cur_stream()), "stem_wgrad")
    grad = torch.full((Co, 3, 3, 3), float("nan"), device="cuda")
    L.check(lib.mnas_wgrad_finalize(partial.data_ptr(), nparts, Co, 27, 1, grad.data_ptr(), 0, L.cur_stream()))
    assert relerr(grad.cpu(), ref_dw) < TOL_F32
    # input gradient (dL/d image, fp32 NCHW): bf16 dy and weights, fp32 accumulate; with the fused input normalisation's scale
    ref_dx = torch.nn.grad.conv2d_input((N, 3, H, W), w, bf16r(dy), stride=2, padding=1)
    w32 = w.cuda().contiguous()
    for aff in (None, torch.tensor([[2.0, 0.5, 1.25], [0.1, 0.2, 0.3]])):
        dx = torch.full((N, 3, H, W), float("nan"), device="cuda")
        affd = aff.cuda().contiguous() if aff is not None else None
        gi = grad_in(gd, yd, bd)
        L.check(lib.mnas_stem_dgrad(C.byref(gi), w32.data_ptr(), N, H, W, Ho, Wo, Co, affd.data_ptr() if aff is not None else None,
                                    dx.data_ptr(), L.cur_stream()), "stem_dgrad")
        want = ref_dx if aff is None else ref_dx * aff[0].view(1, 3, 1, 1)
        assert relerr(dx.cpu(), want) < TOL_F32
    assert lib.mnas_stem_dgrad(C.byref(gi), w32.data_ptr(), N, H, W, Ho + 1, Wo, Co, None, dx.data_ptr(), L.cur_stream()) == L.EINVAL


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C_,rows", [(16, 1000), (48, 777), (72, 301), (1152, 98), (240, 1570)])
def test_bn_bwd(C_, rows):
    lib = L.load()
    g, y = _x((rows, C_), 1), _x((rows, C_), 2)
    b = rand_bn_coefs(C_, 9, O)
    gd, yd, bd = g.to(torch.bfloat16).cuda(), y.to(torch.bfloat16).cuda(), b.clone().cuda()
    nparts = 7
    partial = torch.full((2, C_, nparts), float("nan"), device="cuda")
    L.check(lib.mnas_bn_bwd_reduce(gd.data_ptr(), yd.data_ptr(), bd.data_ptr(), rows, C_, nparts, partial.data_ptr(),
                                   L.cur_stream()))
    s, t, mean, invstd = b[0], b[1], b[5], b[6]
    dz = (g * ((s * y + t) > 0)).double()
    xhat = ((y - mean) * invstd).double()
    S1, S2 = dz.sum(0), (dz * xhat).sum(0)
    p = partial.cpu().double().sum(-1)
    assert relerr(p[0], S1) < 1e-4 and relerr(p[1], S2) < 1e-4
    dgamma = torch.full((C_,), 2.0, device="cuda")
    dbeta = torch.full((C_,), 3.0, device="cuda")
    L.check(lib.mnas_bn_bwd_finalize(partial.data_ptr(), nparts, C_, float(rows), bd.data_ptr(), dgamma.data_ptr(),
                                     dbeta.data_ptr(), 1, L.cur_stream()))
    assert relerr(dgamma.cpu() - 2.0, S2) < 1e-4 and relerr(dbeta.cpu() - 3.0, S1) < 1e-4
    out = bd.cpu().double()
    sd = s.double()
    assert relerr(out[2], sd) < 1e-6
    assert relerr(out[3], -sd * invstd.double() * S2 / rows) < 1e-4
    assert relerr(out[4], sd * (mean.double() * invstd.double() * S2 / rows - S1 / rows)) < 1e-4
    # the coefficients reproduce native_batch_norm_backward's dy
    dy_formula = sd * invstd.double() / sd * 0 + (sd * (dz - S1 / rows - xhat * S2 / rows))
    dy_coef = out[2] * dz + out[3] * y.double() + out[4]
    assert relerr(dy_coef, dy_formula) < 1e-4


@pytest.mark.parametrize("C_,rows,HW", [(16, 500, 50), (320, 98, 49), (96, 392, 196)])
def test_add_act_and_layouts(C_, rows, HW):
    lib = L.load()
    a, b = _x((rows, C_), 1), _x((rows, C_), 2)
    sa, ta = 1 + 0.3 * O.det_uniform((C_,), 4), 0.2 * O.det_uniform((C_,), 5)
    sb, tb = 1 + 0.3 * O.det_uniform((C_,), 6), 0.2 * O.det_uniform((C_,), 7)
    ad, bd = a.to(torch.bfloat16).cuda(), b.to(torch.bfloat16).cuda()
    dsa, dta, dsb, dtb = sa.cuda(), ta.cuda(), sb.cuda(), tb.cuda()
    ref = F.relu(a * sa + ta) + F.relu(b * sb + tb)
    out = torch.empty((rows, C_), dtype=torch.bfloat16, device="cuda")
    N = rows // HW
    onchw = torch.full((N, C_, HW), float("nan"), device="cuda")
    A, B = act_in(ad, dsa, dta), act_in(bd, dsb, dtb)
    L.check(lib.mnas_add_act(C.byref(A), C.byref(B), rows, C_, out.data_ptr(), onchw.data_ptr(), HW, L.cur_stream()))
    assert relerr(out.float().cpu(), ref) < TOL_BF16
    assert relerr(onchw.cpu(), ref.view(N, HW, C_).permute(0, 2, 1)) < 1e-6
    # identity + single input
    A2 = act_in(ad)
    L.check(lib.mnas_add_act(C.byref(A2), None, rows, C_, out.data_ptr(), 0, HW, L.cur_stream()))
    assert relerr(out.float().cpu(), a) == 0.0
    # incoming gradient conversion
    src = O.det_uniform((N, C_, HW), 8)
    dst = torch.empty((N, HW, C_), dtype=torch.bfloat16, device="cuda")
    sd = src.cuda()
    L.check(lib.mnas_nchw_f32_to_nhwc_bf16(sd.data_ptr(), dst.data_ptr(), N, C_, HW, L.cur_stream()))
    assert relerr(dst.float().cpu(), bf16r(src).permute(0, 2, 1)) == 0.0


def test_adam_matches_torch():
    lib = L.load()
    n = 10007
    p0, g0 = O.det_uniform((n,), 1), O.det_uniform((n,), 2)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p], lr=1e-3)
    pd, m, v = p0.clone().cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 4):
        g = g0 * step
        p.grad = g.clone()
        opt.step()
        gd = g.cuda()
        L.check(lib.mnas_adam_step(pd.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8,
                                   0.0, step, 1.0, L.cur_stream()))
    assert relerr(pd.cpu(), p.detach()) < 1e-5


def test_run_ops_batch():
    """The batched launcher reproduces the individual calls (pack -> pw fwd -> bn finalize)."""
    lib = L.load()
    N, H, W, Ci, Co = 2, 12, 12, 16, 48
    x = _x((N, Ci, H, W), 1)
    w = bf16r(O.det_param("t.conv.weight", (Co, Ci, 1, 1), 2))
    xd, wd = nhwc(x), w.cuda().contiguous()
    wp = torch.empty(lib.mnas_packed_bytes(L.PACK_FWD, Co, Ci, 1, 1), dtype=torch.uint8, device="cuda")
    out = torch.empty((N, H, W, Co), dtype=torch.bfloat16, device="cuda")
    nparts = 4
    st = torch.empty((2, Co, nparts), device="cuda")
    gamma, beta = torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda")
    rm, rv = torch.zeros(Co, device="cuda"), torch.ones(Co, device="cuda")
    bn = torch.zeros(8, Co, device="cuda")
    ops = (L.MnasOp * 3)()
    ops[0].opcode = L.OP_PACK_WEIGHTS
    ops[0].i[0:5] = [L.PACK_FWD, Co, Ci, 1, 1]
    ops[0].p[0], ops[0].p[1] = wd.data_ptr(), wp.data_ptr()
    ops[1].opcode = L.OP_CONV_GEMM
    ops[1].i[0:13] = [0, N, H, W, Ci, H, W, Co, 1, 1, 1, 0, nparts]
    ops[1].p[0], ops[1].p[6], ops[1].p[9], ops[1].p[10] = xd.data_ptr(), wp.data_ptr(), out.data_ptr(), st.data_ptr()
    ops[2].opcode = L.OP_BN_FWD_FINALIZE
    ops[2].i[0:3] = [nparts, Co, 1]
    ops[2].d[0], ops[2].d[1], ops[2].d[2] = float(N * H * W), 0.1, 1e-5
    for j, t in enumerate((st, gamma, beta, rm, rv)):
        ops[2].p[j] = t.data_ptr()
    ops[2].p[5], ops[2].p[6] = None, bn.data_ptr()
    failed = C.c_int(-1)
    L.check(lib.mnas_run_ops(ops, 3, L.cur_stream(), C.byref(failed)), "run_ops")
    ref = F.conv2d(x, w)
    assert relerr(from_nhwc(out), ref) < TOL_BF16
    mean = ref.double().mean((0, 2, 3))
    assert relerr(bn[5].cpu(), mean) < TOL_F32
    # a bad opcode is reported, not ignored
    ops[1].opcode = 999
    assert lib.mnas_run_ops(ops, 3, L.cur_stream(), C.byref(failed)) != 0 and failed.value == 1


@pytest.mark.parametrize("C_,nparts", [(16, 1024), (40, 257), (1152, 64), (1152, 513)])
def test_bn_bwd_finalize_long_rows(C_, nparts):
    """Both finalize shapes (wave per channel / block per channel) against an fp64 sum of the same partial table."""
    lib = L.load()
    partial = O.det_uniform((2, C_, nparts), 31)
    b = rand_bn_coefs(C_, 9, O)
    bd, pd = b.clone().cuda(), partial.clone().cuda()
    dgamma, dbeta = torch.zeros(C_, device="cuda"), torch.zeros(C_, device="cuda")
    rows = 12345.0
    L.check(lib.mnas_bn_bwd_finalize(pd.data_ptr(), nparts, C_, rows, bd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                     0, L.cur_stream()))
    S = partial.double().sum(-1)
    s, mean, invstd = b[0].double(), b[5].double(), b[6].double()
    out = bd.cpu().double()
    assert relerr(dbeta.cpu(), S[0]) < 1e-5 and relerr(dgamma.cpu(), S[1]) < 1e-5
    assert relerr(out[3], -s * invstd * S[1] / rows) < 1e-5
    assert relerr(out[4], s * (mean * invstd * S[1] / rows - S[0] / rows)) < 1e-5


@pytest.mark.parametrize("nsplit,Co,Ci,taps,acc", [(5, 24, 16, 1, 0), (300, 16, 48, 1, 1), (300, 16, 48, 1, 0),
                                                   (1024, 16, 32, 1, 1), (129, 8, 8, 9, 1)])
def test_wgrad_finalize_split_ranges(nsplit, Co, Ci, taps, acc):
    """Split ranges longer than one block's share are cut over grid.y (atomic combine) in accumulate mode and walked by
    one block in overwrite mode; dense [Co][taps*Ci] -> [Co][Ci][taps] and depthwise [taps][C] -> [C][taps] relayouts."""
    lib = L.load()
    part = O.det_uniform((nsplit, Co, taps * Ci), 41)
    init = O.det_uniform((Co, Ci, taps), 42)
    grad = init.clone().cuda()
    L.check(lib.mnas_wgrad_finalize(part.cuda().data_ptr(), nsplit, Co, Ci, taps, grad.data_ptr(), acc, L.cur_stream()))
    ref = part.double().sum(0).view(Co, taps, Ci).permute(0, 2, 1) + (init.double() if acc else 0)
    assert relerr(grad.cpu(), ref) < 1e-5
    k = 3
    dpart = O.det_uniform((nsplit, k * k, Co), 43)
    dinit = O.det_uniform((Co, k * k), 44)
    dgrad = dinit.clone().cuda()
    L.check(lib.mnas_dw_wgrad_finalize(dpart.cuda().data_ptr(), nsplit, Co, k, dgrad.data_ptr(), acc, L.cur_stream()))
    dref = dpart.double().sum(0).t() + (dinit.double() if acc else 0)
    assert relerr(dgrad.cpu(), dref) < 1e-5


# ---------------------------------------------------------------------------------------------------
# fused 1x1 backward (input gradient + weight gradient + BatchNorm-backward reduce of x's producer)
PWB = [  # N,H,W,Ci,Co  -- every supported (cin tiles, cout tiles) pair; ragged pixel counts (tile tails)
    (2, 12, 12, 32, 16), (3, 11, 9, 48, 16), (2, 13, 12, 16, 48), (2, 9, 9, 24, 72), (2, 10, 9, 72, 24),
    (1, 9, 8, 40, 240), (1, 11, 7, 240, 40), (1, 9, 7, 480, 80), (1, 10, 7, 576, 96), (2, 9, 8, 40, 120), (2, 11, 7, 120, 40),
]


@pytest.mark.parametrize("shape", PWB)
@pytest.mark.parametrize("variant", ["virt_red", "plain_resid", "virt"])
@pytest.mark.parametrize("nparts", [1, 3])
def test_pw_bwd_fused(shape, variant, nparts):
    lib = L.load()
    N, H, W, Ci, Co = shape
    assert lib.mnas_pw_bwd_supported(Ci, Co) == 1
    M = N * H * W
    x = _x((N, Ci, H, W), 1)
    virt = variant != "plain_resid"
    bx = rand_bn_coefs(Ci, 22, O)                         # bnbuf of x's producer: rows 0/1 = scale/shift
    a = bf16r(F.relu(x * bx[0].view(1, -1, 1, 1) + bx[1].view(1, -1, 1, 1))) if virt else x
    g, y = _x((N, Co, H, W), 6), _x((N, Co, H, W), 7)
    b = rand_bn_coefs(Co, 9, O)
    dy = dy_ref(g, y, b)
    w = bf16r(O.det_param("t.conv.weight", (Co, Ci, 1, 1), 2))
    resid = _x((N, Ci, H, W), 5)
    ref_gin = F.conv_transpose2d(dy, w) + (resid if variant == "plain_resid" else 0)
    ref_dw = torch.nn.grad.conv2d_weight(a, (Co, Ci, 1, 1), dy)
    xd, gd, yd, bd, bxd, rd = nhwc(x), nhwc(g), nhwc(y), b.cuda(), bx.cuda(), nhwc(resid)
    gin = torch.full((N, H, W, Ci), float("nan"), dtype=torch.bfloat16, device="cuda")
    wpart = torch.full((nparts, Co, Ci), float("nan"), device="cuda")
    redp = torch.full((2, Ci, nparts), float("nan"), device="cuda")
    c = L.MnasPwBwd()
    c.M, c.Ci, c.Co, c.nparts = M, Ci, Co, nparts
    c.x = act_in(xd, bxd[0], bxd[1]) if virt else act_in(xd)
    c.dy = grad_in(gd, yd, bd)
    c.w, c.gin, c.wpartial = L.ptr(pack(w, L.PACK_DGRAD)), L.ptr(gin), L.ptr(wpart)
    c.resid = L.ptr(rd) if variant == "plain_resid" else None
    if variant == "virt_red":
        c.red_partial, c.red_y, c.red_bn = L.ptr(redp), L.ptr(xd), L.ptr(bxd)
    L.check(lib.mnas_pw_bwd(C.byref(c), L.cur_stream()), "pw_bwd")
    assert relerr(from_nhwc(gin), ref_gin) < TOL_BF16
    grad = torch.full((Co, Ci, 1, 1), float("nan"), device="cuda")
    L.check(lib.mnas_wgrad_finalize(wpart.data_ptr(), nparts, Co, Ci, 1, grad.data_ptr(), 0, L.cur_stream()))
    assert relerr(grad.cpu(), ref_dw) < TOL_F32
    if variant == "virt_red":
        gq = from_nhwc(gin)                               # the reduce sees g as stored (bf16)
        s_, t_, mu_, is_ = (bx[i].view(1, -1, 1, 1) for i in (0, 1, 5, 6))
        dz = (gq * ((s_ * x + t_) > 0)).double()
        xhat = (x * is_ - mu_ * is_).double()
        st = redp.cpu().double().sum(-1)
        assert relerr(st[0], dz.sum((0, 2, 3))) < 1e-3
        assert relerr(st[1], (dz * xhat).sum((0, 2, 3))) < 1e-3


def test_pw_bwd_rejects_unsupported():
    lib = L.load()
    assert lib.mnas_pw_bwd_supported(96, 576) == 0 and lib.mnas_pw_bwd_supported(12, 16) == 0
    assert lib.mnas_pw_bwd_supported(576, 96) == 1 and lib.mnas_pw_bwd_supported(480, 80) == 1
    c = L.MnasPwBwd()
    c.M, c.Ci, c.Co, c.nparts = 64, 96, 576, 4
    assert lib.mnas_pw_bwd(C.byref(c), L.cur_stream()) == 10001          # MNAS_EINVAL


def test_pack_weights_batch_matches_single():
    """One batched launch packs what the per-tensor entry point packs (all three kinds, ragged shapes)."""
    lib = L.load()
    cases = [(L.PACK_FWD, 24, 16, 9), (L.PACK_DGRAD, 24, 16, 9), (L.PACK_FWD, 40, 240, 1), (L.PACK_DGRAD, 40, 240, 1),
             (L.PACK_DW, 72, 1, 25), (L.PACK_FWD, 32, 27, 1)]
    ws, singles, batched = [], [], []
    host = (L.MnasPackDesc * len(cases))()
    for n, (kind, Co, Ci, taps) in enumerate(cases):
        k = int(round(taps ** 0.5))
        shape = (Co, Ci, k, k) if k * k == taps else (Co, Ci, 1, taps)
        w = O.det_uniform(shape, 50 + n).cuda().contiguous()
        nbytes = lib.mnas_packed_bytes(kind, Co, Ci, 1, taps)
        a = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        b = torch.full((nbytes,), 0xAB, dtype=torch.uint8, device="cuda")
        L.check(lib.mnas_pack_weights(w.data_ptr(), kind, Co, Ci, 1, taps, a.data_ptr(), L.cur_stream()))
        host[n].w, host[n].dst, host[n].kind, host[n].Co, host[n].Ci, host[n].taps = w.data_ptr(), b.data_ptr(), kind, Co, Ci, taps
        ws.append(w); singles.append(a); batched.append(b)
    dev = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).cuda()
    L.check(lib.mnas_pack_weights_batch(dev.data_ptr(), len(cases), L.cur_stream()))
    torch.cuda.synchronize()
    for a, b in zip(singles, batched):
        assert torch.equal(a, b)
