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
