"""-m gpu: the kernel forms of the spatially tiled fused inverted-residual block (round 4; mnasnet.py:105-137 with the expanded
tensors y1 / g2 kept off HBM on the bandwidth-bound 112x112 / 56x56 maps), through the C ABI.

Every fused form recomputes, bit for bit, what the per-layer kernels stored -- so the oracle here is the per-layer kernel itself
(which tests/test_gpu_kernels.py pins against fp32 CPU math on the same bf16-rounded operands):

* mnas_pw_bwd NOGIN  (gin == NULL): weight-gradient partials and the fused BatchNorm-backward sums `torch.equal` to the plain
  launch; dy_out equals the dy tile the plain launch staged (checked against the fp32 formula, bf16 tolerance).
* mnas_pw_bwd RECOMP (dy.y == NULL, w_fwd): input gradient, weight-gradient partials and fused sums `torch.equal` to the plain
  launch fed with the y that mnas_conv_gemm(mode 0) stored.
* mnas_dw_bwd SRC: input gradient, weight-gradient partials and fused sums `torch.equal` to the plain launch fed with the g2 /
  y1 tensors the per-layer kernels stored."""
import ctypes as C

import pytest
import torch

from cases import O
from gpu_util import L, act_in, bf16r, conv_gemm, dy_ref, from_nhwc, grad_in, nhwc, pack, rand_bn_coefs, relerr

pytestmark = pytest.mark.gpu
TOL_BF16 = 6e-3      # one bf16 rounding of an O(1) value, relative to max|ref| (as tests/test_gpu_kernels.py)


def _x(shape, seed):
    return bf16r(O.det_uniform(shape, seed))


def _pw_bwd(M, Ci, Co, nparts, x, dy, w, gin=None, resid=None, red=None, dy_out=None, w_fwd=None, b_fwd=None):
    lib = L.load()
    wpart = torch.full((nparts, Co, Ci), float("nan"), device="cuda")
    redp = torch.full((2, Ci, nparts), float("nan"), device="cuda") if red is not None else None
    c = L.MnasPwBwd()
    c.M, c.Ci, c.Co, c.nparts = M, Ci, Co, nparts
    c.x, c.dy = x, dy
    c.w, c.gin, c.wpartial, c.resid = L.ptr(w), L.ptr(gin), L.ptr(wpart), L.ptr(resid)
    if red is not None:
        c.red_partial, c.red_y, c.red_bn = L.ptr(redp), L.ptr(red[0]), L.ptr(red[1])
    c.dy_out, c.w_fwd, c.b_fwd = L.ptr(dy_out), L.ptr(w_fwd), L.ptr(b_fwd)
    rc = lib.mnas_pw_bwd(C.byref(c), L.cur_stream())
    torch.cuda.synchronize()
    return rc, wpart, redp


NOGIN = [(2, 12, 12, 48, 16), (3, 11, 9, 48, 16), (2, 10, 9, 72, 24), (1, 37, 29, 48, 16)]


@pytest.mark.parametrize("shape", NOGIN)
@pytest.mark.parametrize("nparts", [1, 5])
def test_pw_bwd_nogin(shape, nparts):
    N, H, W, Ci, Co = shape
    M = N * H * W
    x = _x((N, Ci, H, W), 1)
    bx = rand_bn_coefs(Ci, 22, O)
    g, y = _x((N, Co, H, W), 6), _x((N, Co, H, W), 7)
    b = rand_bn_coefs(Co, 9, O)
    w = bf16r(O.det_param("t.conv.weight", (Co, Ci, 1, 1), 2))
    xd, gd, yd, bd, bxd = nhwc(x), nhwc(g), nhwc(y), b.cuda(), bx.cuda()
    wp = pack(w, L.PACK_DGRAD)
    xin, dyin = act_in(xd, bxd[0], bxd[1]), grad_in(gd, yd, bd)
    gin = torch.full((N, H, W, Ci), float("nan"), dtype=torch.bfloat16, device="cuda")
    rc, wpart0, red0 = _pw_bwd(M, Ci, Co, nparts, xin, dyin, wp, gin=gin, red=(xd, bxd))
    assert rc == 0
    dyo = torch.full((N, H, W, Co), float("nan"), dtype=torch.bfloat16, device="cuda")
    rc, wpart1, red1 = _pw_bwd(M, Ci, Co, nparts, xin, dyin, wp, gin=None, red=(xd, bxd), dy_out=dyo)
    assert rc == 0
    assert torch.equal(wpart0, wpart1), "weight-gradient partials differ from the plain launch"
    assert torch.equal(red0, red1), "fused BatchNorm-backward sums differ from the plain launch"
    assert relerr(from_nhwc(dyo), dy_ref(g, y, b)) < TOL_BF16
    # without the reduce there is nothing left to compute: rejected
    rc, _, _ = _pw_bwd(M, Ci, Co, nparts, xin, dyin, wp, gin=None, red=None)
    assert rc == L.EINVAL


RECOMP = [(2, 13, 12, 16, 48), (2, 9, 9, 24, 72), (1, 37, 29, 16, 48), (3, 8, 8, 24, 72)]


@pytest.mark.parametrize("shape", RECOMP)
@pytest.mark.parametrize("variant", ["virt_resid_red", "plain"])
@pytest.mark.parametrize("nparts", [1, 3])
def test_pw_bwd_recomp(shape, variant, nparts):
    N, H, W, Ci, Co = shape
    M = N * H * W
    x = _x((N, Ci, H, W), 1)
    virt = variant != "plain"
    bx = rand_bn_coefs(Ci, 22, O)
    g = _x((N, Co, H, W), 6)
    b = rand_bn_coefs(Co, 9, O)
    w = bf16r(O.det_param("t.conv.weight", (Co, Ci, 1, 1), 2))
    bias = 0.1 * O.det_uniform((Co,), 3)
    resid = _x((N, Ci, H, W), 5)
    xd, gd, bd, bxd, rd, biasd = nhwc(x), nhwc(g), b.cuda(), bx.cuda(), nhwc(resid), bias.cuda()
    xin = act_in(xd, bxd[0], bxd[1]) if virt else act_in(xd)
    wf, wdg = pack(w, L.PACK_FWD), pack(w, L.PACK_DGRAD)
    # the y the forward kernel stores
    yd, _ = conv_gemm(0, N, H, W, Ci, H, W, Co, 1, 1, 0, wf, bias=biasd, act=xin, nparts=max(1, min(64, M // 64)))
    red = (xd, bxd) if virt else None
    rs = rd if virt else None
    gin0 = torch.full((N, H, W, Ci), float("nan"), dtype=torch.bfloat16, device="cuda")
    rc, wpart0, red0 = _pw_bwd(M, Ci, Co, nparts, xin, grad_in(gd, yd, bd), wdg, gin=gin0, resid=rs, red=red)
    assert rc == 0
    gin1 = torch.full((N, H, W, Ci), float("nan"), dtype=torch.bfloat16, device="cuda")
    rc, wpart1, red1 = _pw_bwd(M, Ci, Co, nparts, xin, grad_in(gd, None, bd), wdg, gin=gin1, resid=rs, red=red, w_fwd=wf, b_fwd=biasd)
    assert rc == 0
    assert torch.equal(gin0.view(torch.int16), gin1.view(torch.int16)), \
        "input gradient differs: max |d| = %g" % float((gin0.float() - gin1.float()).abs().max())
    assert torch.equal(wpart0, wpart1)
    if virt:
        assert torch.equal(red0, red1)
    # dy.y == NULL without forward weights: rejected
    rc, _, _ = _pw_bwd(M, Ci, Co, nparts, xin, grad_in(gd, None, bd), wdg, gin=gin1)
    assert rc == L.EINVAL


# ---------------------------------------------------------------------------------------------------
# depthwise backward, SRC form: g2 and y1 recomputed from the block's narrow tensors
SRC = [  # N,H,W,C,E,k
    (2, 12, 12, 16, 48, 3), (1, 37, 29, 16, 48, 3), (2, 20, 20, 24, 72, 5), (1, 9, 50, 24, 72, 5), (1, 112, 112, 16, 48, 3),
    (2, 56, 56, 24, 72, 5), (3, 5, 7, 16, 48, 3),
]


@pytest.mark.parametrize("shape", SRC)
@pytest.mark.parametrize("virt", [False, True])
def test_dw_bwd_src(shape, virt):
    """Build the tensors the per-layer kernels would have stored (y1 from mnas_conv_gemm mode 0, g2 from mnas_pw_bwd together
    with the materialised dy3), run the plain fused depthwise backward on them, then the SRC form on (x, dy3) only."""
    lib = L.load()
    N, H, W, Cc, E, k = shape
    M = N * H * W
    x = _x((N, Cc, H, W), 1)
    bx = rand_bn_coefs(Cc, 21, O)
    w1 = bf16r(O.det_param("t.e.weight", (E, Cc, 1, 1), 2))
    b1 = 0.1 * O.det_uniform((E,), 3)
    w3 = bf16r(O.det_param("t.p.weight", (Cc, E, 1, 1), 4))
    wd = O.det_param("t.dw.weight", (E, 1, k, k), 6)
    bn1, bn2, bn3 = rand_bn_coefs(E, 31, O), rand_bn_coefs(E, 32, O), rand_bn_coefs(Cc, 33, O)
    y2 = _x((N, E, H, W), 8)            # raw depthwise output (any values: the backward only reads it)
    G, y3 = _x((N, Cc, H, W), 9), _x((N, Cc, H, W), 10)
    xd, y2d, Gd, y3d = nhwc(x), nhwc(y2), nhwc(G), nhwc(y3)
    bxd, bn1d, bn2d, bn3d, b1d = bx.cuda(), bn1.cuda(), bn2.cuda(), bn3.cuda(), b1.cuda()
    xin = act_in(xd, bxd[0], bxd[1]) if virt else act_in(xd)
    w1f, w3d, wdp = pack(w1, L.PACK_FWD), pack(w3, L.PACK_DGRAD), pack(wd, L.PACK_DW)
    # per-layer tensors
    y1d, _ = conv_gemm(0, N, H, W, Cc, H, W, E, 1, 1, 0, w1f, bias=b1d, act=xin, nparts=max(1, min(64, M // 64)))
    g2d = torch.full((N, H, W, E), float("nan"), dtype=torch.bfloat16, device="cuda")
    dy3d = torch.full((N, H, W, Cc), float("nan"), dtype=torch.bfloat16, device="cuda")
    a2in = act_in(y2d, bn2d[0], bn2d[1])
    rc, _, _ = _pw_bwd(M, E, Cc, 3, a2in, grad_in(Gd, y3d, bn3d), w3d, gin=g2d, red=(y2d, bn2d))
    assert rc == 0
    rc, _, _ = _pw_bwd(M, E, Cc, 3, a2in, grad_in(Gd, y3d, bn3d), w3d, gin=None, red=(y2d, bn2d), dy_out=dy3d)
    assert rc == 0
    nparts = 37

    def run(src):
        rows = lib.mnas_dw_src_rows(N, H, W, E, k, Cc, nparts) if src else lib.mnas_dw_rows(N, H, W, E, k, nparts, 1)
        assert rows >= 1
        gin = torch.full((N, H, W, E), float("nan"), dtype=torch.bfloat16, device="cuda")
        wpart = torch.full((rows, k * k, E), float("nan"), device="cuda")
        redp = torch.full((2, E, rows), float("nan"), device="cuda")
        a_ = L.MnasDwBwd()
        a_.N, a_.H, a_.W, a_.C, a_.k, a_.nparts, a_.phase = N, H, W, E, k, nparts, 0
        a_.w, a_.gin, a_.wpartial = wdp.data_ptr(), gin.data_ptr(), wpart.data_ptr()
        a_.red_bn, a_.red_partial = bn1d.data_ptr(), redp.data_ptr()
        if src:
            a_.x, a_.dy = act_in(None, bn1d[0], bn1d[1]), grad_in(None, y2d, bn2d)
            a_.src_cin, a_.src_x = Cc, xin
            a_.src_w1, a_.src_b1, a_.src_dy, a_.src_w3t = w1f.data_ptr(), b1d.data_ptr(), dy3d.data_ptr(), w3d.data_ptr()
        else:
            a_.x, a_.dy = act_in(y1d, bn1d[0], bn1d[1]), grad_in(g2d, y2d, bn2d)
        L.check(lib.mnas_dw_bwd(C.byref(a_), L.cur_stream()), "dw_bwd")
        torch.cuda.synchronize()
        grad = torch.full((E, 1, k, k), float("nan"), device="cuda")
        L.check(lib.mnas_dw_wgrad_finalize(wpart.data_ptr(), rows, E, k, grad.data_ptr(), 0, L.cur_stream()))
        return gin, grad, redp.double().sum(-1)

    g_ref, w_ref, r_ref = run(False)
    g_src, w_src, r_src = run(True)
    assert torch.equal(g_ref.view(torch.int16), g_src.view(torch.int16)), \
        "input gradient differs: max |d| = %g" % float((g_ref.float() - g_src.float()).abs().max())
    # different strip geometry = different summation order of the partial tables
    assert relerr(w_src, w_ref) < 1e-4 and relerr(r_src, r_ref) < 1e-4


# ---------------------------------------------------------------------------------------------------
# masked gradient hand-over: mnas_pw_bwd(gin_masked) -> mnas_dw_bwd(g_masked)
MASKED = [(2, 12, 12, 16, 48, 3), (2, 20, 20, 24, 72, 5), (1, 28, 28, 40, 240, 5), (2, 14, 14, 96, 576, 5), (1, 14, 14, 80, 480, 3)]


@pytest.mark.parametrize("shape", MASKED)
def test_masked_gradient_handover(shape):
    """A project conv's fused backward with gin_masked stores dz = g*[s*y2+t>0] (exactly g or 0 in bf16); the depthwise
    backward with g_masked on that tensor must reproduce the plain pair (unmasked g, mask derived on read) bit for bit."""
    lib = L.load()
    N, H, W, Cc, E, k = shape
    M = N * H * W
    assert lib.mnas_pw_bwd_forms(E, Cc) & 4
    y1, y2 = _x((N, E, H, W), 7), _x((N, E, H, W), 8)
    G, y3 = _x((N, Cc, H, W), 9), _x((N, Cc, H, W), 10)
    w3 = bf16r(O.det_param("t.p.weight", (Cc, E, 1, 1), 4))
    wd = O.det_param("t.dw.weight", (E, 1, k, k), 6)
    bn1, bn2, bn3 = rand_bn_coefs(E, 31, O), rand_bn_coefs(E, 32, O), rand_bn_coefs(Cc, 33, O)
    y1d, y2d, Gd, y3d = nhwc(y1), nhwc(y2), nhwc(G), nhwc(y3)
    bn1d, bn2d, bn3d = bn1.cuda(), bn2.cuda(), bn3.cuda()
    w3d, wdp = pack(w3, L.PACK_DGRAD), pack(wd, L.PACK_DW)
    a2in = act_in(y2d, bn2d[0], bn2d[1])

    def proj(masked):
        gin = torch.full((N, H, W, E), float("nan"), dtype=torch.bfloat16, device="cuda")
        wpart = torch.full((3, Cc, E), float("nan"), device="cuda")
        redp = torch.full((2, E, 3), float("nan"), device="cuda")
        c = L.MnasPwBwd()
        c.M, c.Ci, c.Co, c.nparts = M, E, Cc, 3
        c.x, c.dy = a2in, grad_in(Gd, y3d, bn3d)
        c.w, c.gin, c.wpartial = L.ptr(w3d), L.ptr(gin), L.ptr(wpart)
        c.red_partial, c.red_y, c.red_bn = L.ptr(redp), L.ptr(y2d), L.ptr(bn2d)
        c.gin_masked = 1 if masked else 0
        L.check(lib.mnas_pw_bwd(C.byref(c), L.cur_stream()), "pw_bwd")
        torch.cuda.synchronize()
        return gin, wpart, redp

    g_plain, wp0, rp0 = proj(False)
    g_mask, wp1, rp1 = proj(True)
    assert torch.equal(wp0, wp1) and torch.equal(rp0, rp1)
    s2, t2 = bn2[0].view(1, -1, 1, 1), bn2[1].view(1, -1, 1, 1)
    want = from_nhwc(g_plain) * ((s2 * y2 + t2) > 0)
    assert torch.equal(from_nhwc(g_mask), want), "masked gradient is not g * [s*y+t > 0]"

    def dw(g, masked):
        nparts = 37
        rows = lib.mnas_dw_rows(N, H, W, E, k, nparts, 1)
        gin = torch.full((N, H, W, E), float("nan"), dtype=torch.bfloat16, device="cuda")
        wpart = torch.full((rows, k * k, E), float("nan"), device="cuda")
        redp = torch.full((2, E, rows), float("nan"), device="cuda")
        a_ = L.MnasDwBwd()
        a_.N, a_.H, a_.W, a_.C, a_.k, a_.nparts, a_.phase = N, H, W, E, k, nparts, 0
        a_.x, a_.dy = act_in(y1d, bn1d[0], bn1d[1]), grad_in(g, y2d, bn2d)
        a_.w, a_.gin, a_.wpartial = wdp.data_ptr(), gin.data_ptr(), wpart.data_ptr()
        a_.red_bn, a_.red_partial = bn1d.data_ptr(), redp.data_ptr()
        a_.g_masked = 1 if masked else 0
        L.check(lib.mnas_dw_bwd(C.byref(a_), L.cur_stream()), "dw_bwd")
        torch.cuda.synchronize()
        return gin, wpart, redp

    r0, r1 = dw(g_plain, False), dw(g_mask, True)
    for a, b in zip(r0, r1):
        assert torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a, b.view(torch.int16) if b.dtype == torch.bfloat16 else b)
    # the masked flag without the fused reduce is rejected
    a_ = L.MnasDwBwd()
    a_.N, a_.H, a_.W, a_.C, a_.k, a_.nparts, a_.phase = N, H, W, E, k, 37, 1
    a_.x, a_.dy = act_in(y1d, bn1d[0], bn1d[1]), grad_in(g_mask, y2d, bn2d)
    a_.w, a_.gin = wdp.data_ptr(), g_plain.data_ptr()
    a_.g_masked = 1
    assert lib.mnas_dw_bwd(C.byref(a_), L.cur_stream()) == L.EINVAL
