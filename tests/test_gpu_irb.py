"""-m gpu: the fused inverted-residual-block kernels (csrc/mnas_irb.hip, mnas_irb_bwd.hip; MBConv_block, mnasnet.py:105-137 on the
14x14 / 7x7 stages) through the C ABI against fp32 CPU math on the same bf16-rounded operands.  Tolerances as in
test_gpu_kernels.py: bf16 outputs <= 6e-3 of max |ref|, fp32 reductions <= 2e-3."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from cases import O
from gpu_util import L, act_in, bf16r, from_nhwc, nhwc, pack, relerr

pytestmark = pytest.mark.gpu

# (N, H, W, C, E, k): the three block shapes of MNASNet-1.0's small-map stages (fewer images), a 5x5 7x7 variant, odd N
IRB_SHAPES = [(8, 14, 14, 80, 480, 3), (9, 14, 14, 96, 576, 5), (7, 7, 7, 192, 1152, 3), (6, 7, 7, 192, 1152, 5),
              (3, 14, 14, 16, 64, 5), (4, 10, 14, 24, 96, 3)]


def _block_inputs(N, H, W, Cc, E, k, seed, virt):
    x = bf16r(O.det_uniform((N, Cc, H, W), seed))
    xs = xt = None
    if virt:
        xs = 1.0 + 0.3 * O.det_uniform((Cc,), seed + 1)
        xt = 0.2 * O.det_uniform((Cc,), seed + 2)
    w1 = O.det_uniform((E, Cc, 1, 1), seed + 3) * (1.5 / Cc ** 0.5)
    b1 = 0.1 * O.det_uniform((E,), seed + 4)
    bn1 = torch.zeros(8, E)
    bn1[0] = 1.0 + 0.3 * O.det_uniform((E,), seed + 5)
    bn1[1] = 0.2 * O.det_uniform((E,), seed + 6)
    wdw = O.det_uniform((E, 1, k, k), seed + 7) * (1.0 / k)
    bdw = 0.1 * O.det_uniform((E,), seed + 8)
    return x, xs, xt, w1, b1, bn1, wdw, bdw


def _fwd_ref(x, xs, xt, w1, b1, bn1, wdw, bdw, k):
    a0 = x if xs is None else bf16r(torch.relu(x * xs.view(1, -1, 1, 1) + xt.view(1, -1, 1, 1)))
    y1f = F.conv2d(a0, bf16r(w1)) + b1.view(1, -1, 1, 1)
    y1 = bf16r(y1f)
    a1 = bf16r(torch.relu(y1 * bn1[0].view(1, -1, 1, 1) + bn1[1].view(1, -1, 1, 1)))
    y2 = F.conv2d(a1, wdw, padding=k // 2, groups=wdw.shape[0]) + bdw.view(1, -1, 1, 1)
    return a0, y1, a1, y2


@pytest.mark.parametrize("virt", [False, True])
@pytest.mark.parametrize("shape", IRB_SHAPES)
def test_irb_fwd(shape, virt):
    N, H, W, Cc, E, k = shape
    lib = L.load()
    assert lib.mnas_irb_supported(N, H, W, Cc, E, k) == 1
    x, xs, xt, w1, b1, bn1, wdw, bdw = _block_inputs(N, H, W, Cc, E, k, 300, virt)
    _, y1r, _, y2r = _fwd_ref(x, xs, xt, w1, b1, bn1, wdw, bdw, k)
    for want, store_y1 in ((3, True), (64, False)):
        nparts = lib.mnas_irb_fwd_parts(N, H, W, Cc, E, k, want)
        assert 1 <= nparts <= want
        xd = nhwc(x)
        sd, td = (xs.cuda(), xt.cuda()) if virt else (None, None)
        w1p, wdp = pack(w1, L.PACK_FWD), pack(wdw, L.PACK_DW)
        b1d, bn1d, bdd = b1.cuda(), bn1.cuda().contiguous(), bdw.cuda()
        y1 = torch.full((N, H, W, E), float("nan"), dtype=torch.bfloat16, device="cuda") if store_y1 else None
        y2 = torch.full((N, H, W, E), float("nan"), dtype=torch.bfloat16, device="cuda")
        st = torch.full((2, E, nparts), float("nan"), device="cuda")
        a = L.MnasIrbFwd()
        a.N, a.H, a.W, a.C, a.E, a.k, a.nparts = N, H, W, Cc, E, k, nparts
        a.x = act_in(xd, sd, td)
        a.w1, a.b1, a.bn1, a.wdw, a.bdw = L.ptr(w1p), L.ptr(b1d), L.ptr(bn1d), L.ptr(wdp), L.ptr(bdd)
        a.y1, a.y2, a.stats = L.ptr(y1), L.ptr(y2), L.ptr(st)
        L.check(lib.mnas_irb_fwd(C.byref(a), L.cur_stream()), "irb_fwd")
        torch.cuda.synchronize()
        if store_y1:
            assert relerr(from_nhwc(y1), y1r) < 6e-3
        assert relerr(from_nhwc(y2), y2r) < 6e-3, (shape, virt, want)
        s = st.sum(dim=2).cpu()
        assert relerr(s[0], y2r.sum(dim=(0, 2, 3))) < 2e-3
        assert relerr(s[1], (y2r * y2r).sum(dim=(0, 2, 3))) < 2e-3


def test_irb_rejects_unsupported():
    lib = L.load()
    assert lib.mnas_irb_supported(8, 28, 28, 40, 240, 5) == 0        # 28x28 maps stay on the per-layer kernels
    assert lib.mnas_irb_supported(8, 14, 14, 80, 488, 3) == 0        # E % 32
    assert lib.mnas_irb_supported(8, 14, 14, 80, 480, 7) == 0
    a = L.MnasIrbFwd()
    a.N, a.H, a.W, a.C, a.E, a.k, a.nparts = 8, 28, 28, 40, 240, 5, 4
    assert lib.mnas_irb_fwd(C.byref(a), L.cur_stream()) == L.EINVAL


# ---- backward ---------------------------------------------------------------------------------------------------------------
def _mat(t):
    """NCHW -> [N*H*W][C] fp32"""
    return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()


def _bn_rows(E, seed):
    """bnbuf float[8][E] with plausible rows: s, t, c1, c2, c3, mean, invstd"""
    u = O.det_uniform((8, E), seed)
    b = torch.zeros(8, E)
    b[0] = 1.0 + 0.3 * u[0]
    b[1] = 0.2 * u[1]
    b[2] = b[0]
    b[3] = 0.05 * u[3]
    b[4] = 0.02 * u[4]
    b[5] = 0.1 * u[5]
    b[6] = 1.0 + 0.2 * u[6].abs()
    return b


def _irb_desc(N, H, W, Cc, E, k, nparts):
    a = L.MnasIrbBwd()
    a.N, a.H, a.W, a.C, a.E, a.k, a.nparts = N, H, W, Cc, E, k, nparts
    return a


@pytest.mark.parametrize("want", [3, 64])
@pytest.mark.parametrize("shape", IRB_SHAPES)
def test_irb_bwd_proj(shape, want):
    N, H, W, Cc, E, k = shape
    lib = L.load()
    nparts = lib.mnas_irb_fwd_parts(N, H, W, Cc, E, k, want)
    G = bf16r(O.det_uniform((N, Cc, H, W), 410))
    y3 = bf16r(O.det_uniform((N, Cc, H, W), 411))
    y2 = bf16r(O.det_uniform((N, E, H, W), 412))
    bn3, bn2 = _bn_rows(Cc, 413), _bn_rows(E, 414)
    w3 = O.det_uniform((Cc, E, 1, 1), 415) * (1.5 / E ** 0.5)
    # reference
    v = lambda r, b: b[r].view(1, -1, 1, 1)
    dy3 = bf16r(v(2, bn3) * (G * ((v(0, bn3) * y3 + v(1, bn3)) > 0)) + v(3, bn3) * y3 + v(4, bn3))
    w3b = bf16r(w3).view(Cc, E)
    da2 = bf16r(_mat(dy3) @ w3b)                                  # [M][E]
    y2m = _mat(y2)
    dz2 = da2 * ((bn2[0] * y2m + bn2[1]) > 0)
    red_ref = torch.stack([dz2.sum(0), (dz2 * (y2m - bn2[5]) * bn2[6]).sum(0)])
    a2 = bf16r(torch.relu(bn2[0] * y2m + bn2[1]))
    dw3_ref = _mat(dy3).t() @ a2                                  # [C][E]
    # kernel
    a = _irb_desc(N, H, W, Cc, E, k, nparts)
    Gd, y3d, y2d, bn3d, bn2d = nhwc(G), nhwc(y3), nhwc(y2), bn3.cuda().contiguous(), bn2.cuda().contiguous()
    w3t = pack(w3, L.PACK_DGRAD)
    dy3d = torch.full((N, H, W, Cc), float("nan"), dtype=torch.bfloat16, device="cuda")
    wpart = torch.full((nparts, Cc, E), float("nan"), device="cuda")
    red = torch.full((2, E, nparts), float("nan"), device="cuda")
    a.gout = L.MnasGradIn(L.ptr(Gd), L.ptr(y3d), L.ptr(bn3d))
    a.y2, a.bn2, a.w3t = L.ptr(y2d), L.ptr(bn2d), L.ptr(w3t)
    a.dy3, a.w3partial, a.red2 = L.ptr(dy3d), L.ptr(wpart), L.ptr(red)
    L.check(lib.mnas_irb_bwd_proj(C.byref(a), L.cur_stream()), "irb_bwd_proj")
    torch.cuda.synchronize()
    assert relerr(from_nhwc(dy3d), dy3) < 6e-3
    assert relerr(wpart.sum(0).cpu(), dw3_ref) < 3e-3
    r = red.sum(2).cpu()
    assert relerr(r[0], red_ref[0]) < 5e-3 and relerr(r[1], red_ref[1]) < 5e-3


@pytest.mark.parametrize("virt", [False, True])
@pytest.mark.parametrize("shape", IRB_SHAPES)
def test_irb_bwd_dw(shape, virt):
    N, H, W, Cc, E, k = shape
    lib = L.load()
    nparts = lib.mnas_irb_fwd_parts(N, H, W, Cc, E, k, 5)
    x, xs, xt, w1, b1, bn1, wdw, bdw = _block_inputs(N, H, W, Cc, E, k, 500, virt)
    bn1[5] = 0.1 * O.det_uniform((E,), 520)
    bn1[6] = 1.0 + 0.2 * O.det_uniform((E,), 521).abs()
    bn2 = _bn_rows(E, 522)
    dy3 = bf16r(0.5 * O.det_uniform((N, Cc, H, W), 523))
    y2 = bf16r(O.det_uniform((N, E, H, W), 524))
    w3 = O.det_uniform((Cc, E, 1, 1), 525) * (1.5 / E ** 0.5)
    # reference
    a0, y1, a1, _ = _fwd_ref(x, xs, xt, w1, b1, bn1, wdw, bdw, k)
    v = lambda r, b: b[r].view(1, -1, 1, 1)
    da2 = bf16r(_mat(dy3) @ bf16r(w3).view(Cc, E)).view(N, H, W, E).permute(0, 3, 1, 2)
    dz2 = da2 * ((v(0, bn2) * y2 + v(1, bn2)) > 0)
    dy2 = bf16r(v(2, bn2) * dz2 + v(3, bn2) * y2 + v(4, bn2))
    da1 = F.conv_transpose2d(dy2, wdw, padding=k // 2, groups=E)
    dz1 = bf16r(da1) * (a1 > 0)
    dwdw_ref = torch.nn.grad.conv2d_weight(a1, wdw.shape, dy2, padding=k // 2, groups=E)        # [E][1][k][k]
    P_ref = _mat(dz1).t() @ _mat(a0)                               # [E][C]
    y1m, dz1m = _mat(y1), _mat(dz1)
    red_ref = torch.stack([dz1m.sum(0), (dz1m * (y1m - bn1[5]) * bn1[6]).sum(0)])
    # kernel
    a = _irb_desc(N, H, W, Cc, E, k, nparts)
    xd = nhwc(x)
    sd, td = (xs.cuda(), xt.cuda()) if virt else (None, None)
    keep = [xd, sd, td, nhwc(dy3), nhwc(y2), pack(w1, L.PACK_FWD), pack(w3, L.PACK_DGRAD), b1.cuda(), bn1.cuda().contiguous(),
            bn2.cuda().contiguous(), pack(wdw, L.PACK_DW)]
    a.x = act_in(xd, sd, td)
    a.dy3, a.y2, a.w1, a.w3t, a.b1, a.bn1, a.bn2, a.wdw = (L.ptr(t) for t in keep[3:])
    g1 = torch.full((N, H, W, E), float("nan"), dtype=torch.bfloat16, device="cuda")
    dwp = torch.full((nparts, k * k, E), float("nan"), device="cuda")
    pp = torch.full((nparts, E, Cc), float("nan"), device="cuda")
    red = torch.full((2, E, nparts), float("nan"), device="cuda")
    a.g1, a.dwpartial, a.ppartial, a.red1 = L.ptr(g1), L.ptr(dwp), L.ptr(pp), L.ptr(red)
    L.check(lib.mnas_irb_bwd_dw(C.byref(a), L.cur_stream()), "irb_bwd_dw")
    torch.cuda.synchronize()
    # dy2 is staged as bf16 and a ReLU mask can flip on a rounding boundary: compare g1 by relative L2
    g1c = from_nhwc(g1)
    assert float((g1c - dz1).norm() / dz1.norm()) < 1e-2, (shape, virt)
    dwk = dwp.sum(0).cpu().view(k, k, E).permute(2, 0, 1)
    assert relerr(dwk, dwdw_ref.view(E, k, k)) < 5e-3
    assert relerr(pp.sum(0).cpu(), P_ref) < 5e-3
    r = red.sum(2).cpu()
    assert relerr(r[0], red_ref[0]) < 5e-3 and relerr(r[1], red_ref[1]) < 5e-3


@pytest.mark.parametrize("virt,resid", [(False, True), (True, False)])
@pytest.mark.parametrize("shape", IRB_SHAPES)
def test_irb_bwd_exp(shape, virt, resid):
    N, H, W, Cc, E, k = shape
    lib = L.load()
    x, xs, xt, w1, b1, bn1, wdw, bdw = _block_inputs(N, H, W, Cc, E, k, 600, virt)
    bn1[2] = 1.0 + 0.3 * O.det_uniform((E,), 620)
    bn1[3] = 0.05 * O.det_uniform((E,), 621)
    bn1[4] = 0.02 * O.det_uniform((E,), 622)
    dz1 = bf16r(O.det_uniform((N, E, H, W), 623)) * (O.det_uniform((N, E, H, W), 624) > 0)
    G = bf16r(O.det_uniform((N, Cc, H, W), 625)) if resid else None
    a0, y1, _, _ = _fwd_ref(x, xs, xt, w1, b1, bn1, wdw, bdw, k)
    v = lambda r, b: b[r].view(1, -1, 1, 1)
    dy1 = bf16r(v(2, bn1) * dz1 + v(3, bn1) * y1 + v(4, bn1))
    dx = _mat(dy1) @ bf16r(w1).view(E, Cc)
    if resid:
        dx = dx + _mat(G)
    dx_ref = dx.view(N, H, W, Cc).permute(0, 3, 1, 2)
    a = _irb_desc(N, H, W, Cc, E, k, 1)
    xd = nhwc(x)
    sd, td = (xs.cuda(), xt.cuda()) if virt else (None, None)
    g1d, w1p, b1d, bn1d = nhwc(dz1), pack(w1, L.PACK_FWD), b1.cuda(), bn1.cuda().contiguous()
    Gd = nhwc(G) if resid else None
    out = torch.full((N, H, W, Cc), float("nan"), dtype=torch.bfloat16, device="cuda")
    a.x = act_in(xd, sd, td)
    a.g1, a.w1, a.b1, a.bn1, a.dx = L.ptr(g1d), L.ptr(w1p), L.ptr(b1d), L.ptr(bn1d), L.ptr(out)
    a.gout = L.MnasGradIn(L.ptr(Gd), 0, 0)
    L.check(lib.mnas_irb_bwd_exp(C.byref(a), L.cur_stream()), "irb_bwd_exp")
    torch.cuda.synchronize()
    assert relerr(from_nhwc(out), dx_ref) < 8e-3, (shape, virt, resid)


def test_irb_w1_finalize_identity():
    """dW1 = dy1^T a with dy1 = c1*dz1 + c2*y1 + c3 (per expanded channel) equals c1*P + c2*(W1 G + b1 Sx^T) + c3*Sx: the kernel's
    formula against the direct product, on real tensors (P, G, Sx built on the CPU in fp64)."""
    lib = L.load()
    M, Cc, E, nparts = 500, 24, 96, 3
    a0 = bf16r(O.det_uniform((M, Cc), 700)).double()
    w1 = O.det_uniform((E, Cc), 701) * 0.3
    b1 = 0.1 * O.det_uniform((E,), 702)
    bn1 = _bn_rows(E, 703)
    dz1 = (O.det_uniform((M, E), 704) * (O.det_uniform((M, E), 705) > 0)).double()
    y1 = a0 @ bf16r(w1).double().t() + b1.double()
    dy1 = bn1[2].double() * dz1 + bn1[3].double() * y1 + bn1[4].double()
    direct = dy1.t() @ a0                                          # [E][C]
    P = dz1.t() @ a0
    parts = torch.stack([P * f for f in (0.5, 0.3, 0.2)]).float().cuda().contiguous()
    gsum = torch.cat([(a0.t() @ a0).reshape(-1), a0.sum(0)]).cuda().contiguous()
    grad = torch.ones(E, Cc, device="cuda")
    w1d, b1d, bn1d = w1.cuda().contiguous(), b1.cuda(), bn1.cuda().contiguous()
    L.check(lib.mnas_irb_w1_finalize(parts.data_ptr(), nparts, E, Cc, gsum.data_ptr(), w1d.data_ptr(), b1d.data_ptr(),
                                     bn1d.data_ptr(), grad.data_ptr(), 1, L.cur_stream()), "irb_w1_finalize")
    torch.cuda.synchronize()
    assert relerr(grad.cpu() - 1.0, direct.float()) < 1e-4
