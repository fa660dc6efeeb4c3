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
