"""-m gpu: the drop-in modules (mnasnet_pytorch_amd, HIP engine) against
  (a) oracle/bf16_mirror.py -- the reference's arithmetic with bf16 rounding at the HIP path's storage points:
      TIGHT bound: relative L2 <= 2e-3 per tensor for one ConvBlock (measured <= 2e-4: the engine is
      bit-for-bit the mirror up to fp32 summation order), <= 3e-2 for blocks/stages (a 1-ulp bf16 difference
      in y moves the heavily cancelling sum dgamma = sum dz*xhat by ~2 %), whole networks: outputs <= 3e-2,
      gradients by cosine similarity (min >= 0.8, median >= 0.95) because the 57-layer backward amplifies
      those 1-ulp differences (measured median relative L2 0.18) -- this pins the engine's wiring;
  (b) the golden fixtures captured from the fp32 reference: LOOSE bound, because bf16 storage legitimately
      costs a few % at these tiny batch sizes (ReLU-mask flips; see tests/test_bf16_mirror.py, which holds
      the mirror itself to the same goldens on CPU):
         one ConvBlock   y <= 1e-2, gradients <= 8e-2         MBConv_block / stage   y <= 4e-2, gradients <= 0.2
         whole network, well-conditioned state (gain ~5)      y <= 8e-2
         whole network, default state (gain ~110, measured)   y <= 0.7  (sanity only; any bf16 pipeline is here)
"""
import os
import numpy as np
import pytest
import torch

import cases as C
from cases import O
from oracle import bf16_mirror as M
from test_oracle_golden import block_state, prim_state

pytestmark = pytest.mark.gpu
TIGHT = 2e-3      # one ConvBlock
TIGHT_BLK = 3e-2  # blocks / stages


def rl2(a, b):
    a = torch.as_tensor(np.asarray(a)).double().flatten()
    b = torch.as_tensor(np.asarray(b)).double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def load(name):
    return np.load("%s/%s.npz" % (C.GOLDEN_DIR, name))


def fill(module, prefix, seed=C.STATE_SEED):
    """Same closed-form fill as the golden generator (first alias names the value)."""
    sd = module.state_dict()
    new, first = {}, {}
    for k, v in sd.items():
        src = first.setdefault(v.data_ptr(), k) if v.dim() > 0 else k
        new[k] = O.det_param(prefix + "." + src, tuple(v.shape), seed).to(v.dtype)
    module.load_state_dict(new)


def check_grads(module, mirror_grads, prefix_map, tol, gold=None, gold_prefix=None, gold_tol=None):
    worst = 0.0
    for kk, p in module.named_parameters():
        ref = mirror_grads[prefix_map(kk)]
        if kk.endswith("conv.bias"):
            assert p.grad is not None and float(p.grad.abs().max()) < 1e-4
            continue
        e = rl2(p.grad.cpu(), ref)
        worst = max(worst, e)
        # dgamma = sum dz*xhat cancels heavily: a 1-ulp bf16 difference in one y moves it by a few %
        assert e < (max(tol, 0.1) if kk.endswith("bn.weight") and tol > TIGHT else tol), (kk, e)
        if gold is not None:
            assert rl2(p.grad.cpu(), gold[gold_prefix + kk]) < gold_tol, kk
    return worst


@pytest.mark.parametrize("name", sorted(C.PRIMITIVES))
@pytest.mark.parametrize("train", [True, False])
def test_convblock(name, train):
    from mnasnet_pytorch_amd import ConvBlock
    g = load("primitives")
    cin, cout, k, s, p, grp, N, H, W = C.PRIMITIVES[name]
    m = ConvBlock(cin, cout, kernel_size=k, stride=s, padding=p, groups=grp)
    fill(m, name)
    m = m.cuda().train(train)
    x0 = C.det_input((N, cin, H, W))
    x = x0.cuda().requires_grad_(cin != 3)
    y = m(x)
    tag = name + ("/train" if train else "/eval")
    assert y.shape == g[tag + "/y"].shape and y.dtype == torch.float32
    spec = O.ConvSpec("cb", cin, cout, k, s, p, grp)
    st = prim_state(name, spec)
    cot = C.cotangent(tuple(y.shape))
    r = M.run([("conv", spec)], st, x0, train=train, cot=cot if train else None, need_dx=True)
    assert rl2(y.detach().cpu(), r["y"]) < TIGHT
    assert rl2(y.detach().cpu(), g[tag + "/y"]) < 1e-2
    if not train:
        return
    (y * cot.cuda()).sum().backward()
    if cin != 3:
        assert rl2(x.grad.cpu(), r["dx"]) < TIGHT
        assert rl2(x.grad.cpu(), g[tag + "/dx"]) < 8e-2
    check_grads(m, r["grads"], lambda kk: "cb." + kk, TIGHT, g, tag + "/d_", 8e-2)
    assert rl2(m.bn.running_mean.cpu(), g[tag + "/bn.running_mean"]) < 1e-2
    assert rl2(m.bn.running_var.cpu(), g[tag + "/bn.running_var"]) < 1e-2
    assert int(m.bn.num_batches_tracked) == int(g[tag + "/bn.num_batches_tracked"])


@pytest.mark.parametrize("dw_split", [False, True])
@pytest.mark.parametrize("name", sorted(C.BLOCKS))
def test_block(name, dw_split):
    """dw_split=True runs the depthwise backward as two launches (input gradient on the main stream, weight gradient on
    the side stream) instead of the default fused sweep."""
    from mnasnet_pytorch_amd import MBConv_block
    g = load("blocks")
    c, t, k, N, H, W = C.BLOCKS[name]
    m = MBConv_block(c, t, k)
    fill(m, name)
    m = m.cuda().train()
    if dw_split:
        m._engine().dw_fused_k = ()
    x0 = C.det_input((N, c, H, W))
    x = x0.cuda().requires_grad_(True)
    y = m(x)
    cot = C.cotangent(tuple(y.shape))
    (y * cot.cuda()).sum().backward()
    specs = O._block_specs("blk", c, t, k)
    st = block_state(name, specs)
    r = M.run([("block", specs)], st, x0, True, cot, need_dx=True)
    assert rl2(y.detach().cpu(), r["y"]) < TIGHT_BLK
    assert rl2(x.grad.cpu(), r["dx"]) < TIGHT_BLK
    assert rl2(y.detach().cpu(), g[name + "/y"]) < 4e-2
    assert rl2(x.grad.cpu(), g[name + "/dx"]) < 0.3
    check_grads(m, r["grads"], lambda kk: "blk." + kk, TIGHT_BLK, g, name + "/d_", 0.3)
    for kk, v in m.state_dict().items():
        if "running" in kk:
            assert rl2(v.cpu(), g[name + "/" + kk]) < 2e-2, kk


@pytest.mark.parametrize("name", sorted(C.SEPCONVS))
@pytest.mark.parametrize("train", [True, False])
def test_sepconv(name, train):
    """SepConv (mnasnet.py:64-103) as its own parity row: the network's instance, the list-multiplied forms (repeat = 1, 2:
    the shared (depthwise, pointwise) pair is traced `repeat` times with the same parameter pointers -- gradients accumulate,
    the BatchNorm buffers are updated once per application) and reduce=True (STRIDE-2 depthwise convs, mnasnet.py:73-75:
    csrc/mnas_dw2.hip; with repeat = 1 two of them in a row) against the mirror (tight) and the reference's golden (loose)."""
    from mnasnet_pytorch_amd import SepConv
    g = load("sepconvs")
    cin, cout, k, reduce, repeat, N, H, W = C.SEPCONVS[name]
    m = SepConv(cin, cout, kernel_size=k, reduce=reduce, repeat=repeat)
    fill(m, name)
    m = m.cuda().train(train)
    if repeat >= 2:
        assert m.sequence[0] is m.sequence[2] and m.sequence[1] is m.sequence[3]      # list-multiply: ONE module pair
    x0 = C.det_input((N, cin, H, W))
    x = x0.cuda().requires_grad_(True)
    y = m(x)
    tag = name + ("/train" if train else "/eval")
    assert y.shape == g[tag + "/y"].shape
    prog, uniq = C.sepconv_specs(name)
    st = C.sepconv_state(name, uniq)
    cot = C.cotangent(tuple(y.shape))
    r = M.run(prog, st, x0, train=train, cot=cot if train else None, need_dx=True)
    tol = TIGHT_BLK if train else 1e-2
    assert rl2(y.detach().cpu(), r["y"]) < tol
    assert rl2(y.detach().cpu(), g[tag + "/y"]) < 4e-2
    if not train:
        return
    (y * cot.cuda()).sum().backward()
    assert rl2(x.grad.cpu(), r["dx"]) < TIGHT_BLK
    assert rl2(x.grad.cpu(), g[tag + "/dx"]) < 0.3
    check_grads(m, r["grads"], lambda kk: "sep." + kk, TIGHT_BLK, g, tag + "/d_", 0.3)
    sd = m.state_dict()
    for kk in g.files:
        if kk.startswith(tag + "/") and "tracked" in kk:
            assert int(sd[kk[len(tag) + 1:]]) == int(g[kk]), kk                       # `repeat` updates for the shared pair
        if kk.startswith(tag + "/") and "running" in kk:
            assert rl2(sd[kk[len(tag) + 1:]].cpu(), g[kk]) < 4e-2, kk


def _stage_setup(name, proj_gamma, spec=None):
    """Stand-alone stage module + the mirror's program/state (mnasnet.py:139-173).  proj_gamma scales the BatchNorm
    weight of every block's projection ConvBlock (module and mirror alike): 1.0 is the state the goldens were made
    with, 0.1 the well-conditioned variant (same idea as oracle.init_state(proj_gamma=...))."""
    from mnasnet_pytorch_amd import MBConv
    cin, cout, t, layers, k, reduce, ccf, N, H, W = spec if spec is not None else C.STAGES[name]
    m = MBConv(cin, cout, t, layers, kernel_size=k, reduce=reduce, cut_channels_first=ccf)
    fill(m, name)
    stride = 2 if reduce else 1
    bc = cout if ccf else cin
    conv = O.ConvSpec("sequence.%d" % (0 if ccf else layers), cin, cout, 3, stride, 1, 1)
    blk = O._block_specs("sequence.%d" % (1 if ccf else 0), bc, t, k)
    st = {}
    for s_ in [conv] + blk:
        for suf, shp in (("conv.weight", s_.weight_shape()), ("conv.bias", (s_.cout,)), ("bn.weight", (s_.cout,)),
                         ("bn.bias", (s_.cout,)), ("bn.running_mean", (s_.cout,)), ("bn.running_var", (s_.cout,))):
            st[s_.prefix + "." + suf] = O.det_param("%s.%s.%s" % (name, s_.prefix, suf), shp, C.STATE_SEED)
        st[s_.prefix + ".bn.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    if proj_gamma != 1.0:
        st[blk[2].prefix + ".bn.weight"] = st[blk[2].prefix + ".bn.weight"] * proj_gamma
        with torch.no_grad():
            for kk, p in m.named_parameters():          # shared block: one tensor behind every alias
                if kk == blk[2].prefix + ".bn.weight":
                    p.mul_(proj_gamma)
    prog = ([("conv", conv)] if ccf else []) + [("block", blk)] * layers + ([] if ccf else [("conv", conv)])
    return m.cuda().train(), prog, st, (N, cin, H, W)


@pytest.mark.parametrize("name", sorted(C.STAGES))
def test_stage(name):
    """Default state (the goldens' state).  Three applications of one shared, untrained block amplify a 1-ulp bf16
    difference in an early activation into a few % of dx (measured 2-6 % between two summation orders of the SAME
    kernels), so dx is held to the mirror at 0.1 here and TIGHTLY in test_stage_well_conditioned below."""
    g = load("stages")
    m, prog, st, shp = _stage_setup(name, 1.0)
    x0 = C.det_input(shp)
    x = x0.cuda().requires_grad_(True)
    y = m(x)
    cot = C.cotangent(tuple(y.shape))
    (y * cot.cuda()).sum().backward()
    r = M.run(prog, st, x0, True, cot, need_dx=True)
    assert rl2(y.detach().cpu(), r["y"]) < TIGHT_BLK
    assert rl2(x.grad.cpu(), r["dx"]) < 0.1
    assert rl2(y.detach().cpu(), g[name + "/y"]) < 4e-2
    assert rl2(x.grad.cpu(), g[name + "/dx"]) < 0.3
    # shared block: grads are the SUM over its `layers` applications
    check_grads(m, r["grads"], lambda kk: kk, 0.1, g, name + "/d_", 0.3)
    sd = m.state_dict()
    for kk in g.files:
        if kk.startswith(name + "/") and "tracked" in kk:
            assert int(sd[kk[len(name) + 1:]]) == int(g[kk]), kk       # 3 updates per forward for the shared block
        if kk.startswith(name + "/") and "running" in kk:
            assert rl2(sd[kk[len(name) + 1:]].cpu(), g[kk]) < 4e-2, kk


@pytest.mark.parametrize("fused_pw", [False, True])
@pytest.mark.parametrize("name", sorted(C.STAGES))
def test_stage_well_conditioned(name, fused_pw):
    """Projection BatchNorm weights x0.1 (residual branch small against the skip path, gain ~1): here the engine must
    sit on the mirror tightly -- this is the test that pins the stage wiring (shared weights, residuals, stride-2),
    with the separate dgrad/wgrad kernels and with the fused 1x1 backward forced on."""
    m, prog, st, shp = _stage_setup(name, 0.1)
    if fused_pw:
        m._engine().pw_fused_min_pixels = 0
    x0 = C.det_input(shp)
    x = x0.cuda().requires_grad_(True)
    y = m(x)
    cot = C.cotangent(tuple(y.shape))
    (y * cot.cuda()).sum().backward()
    r = M.run(prog, st, x0, True, cot, need_dx=True)
    assert rl2(y.detach().cpu(), r["y"]) < 1e-2
    assert rl2(x.grad.cpu(), r["dx"]) < TIGHT_BLK
    check_grads(m, r["grads"], lambda kk: kk, TIGHT_BLK)


# stages of MNASNet-1.0 (ccf=False) at FULL spatial size and a quarter of the bench batch: the pixel counts at which the
# persistent-grid / multi-tile / channel-slice paths of the kernels engage (fused 1x1 backward with 1024 and 512
# workgroups, 2-row DMA groups with full-width strips, 3x3 stride-2 parity tiling at 112^2 ...)
FULL_STAGES = {
    "features2_16_24_k3_112": (16, 24, 3, 3, 3, True, False, 64, 112, 112),
    "features3_24_40_k5_56": (24, 40, 3, 3, 5, True, False, 64, 56, 56),
    "features4_40_80_k5_28": (40, 80, 6, 3, 5, True, False, 64, 28, 28),
    # ... and at the BASELINE batch (bs 256): the grids the bench launches (M = 802 816 / 200 704 pixels)
    "features3_24_40_k5_56_bs256": (24, 40, 3, 3, 5, True, False, 256, 56, 56),
    "features4_40_80_k5_28_bs256": (40, 80, 6, 3, 5, True, False, 256, 28, 28),
    "features5_80_96_k3_14": (80, 96, 6, 2, 3, False, False, 256, 14, 14),
    "features7_192_320_k3_7": (192, 320, 6, 1, 3, False, False, 256, 7, 7),
    "features6_96_192_k5_14": (96, 192, 6, 4, 5, True, False, 256, 14, 14),
    # the reference's DEFAULT topology, Mnasnet() = cut_channels_first=True (mnasnet.py:176): the stride-2 conv comes first and the
    # blocks run at the OUT width -- other (C, E, k, plane) combinations than everything above (24/72 k3 at 56x56, 40/120 k5 at
    # 28x28, 80/480 k5 and 96/576 k3 at 14x14, 192/1152 k5 and 320/1920 k3 at 7x7), at N >= 64
    "ccfT_features2_16_24_k3_112": (16, 24, 3, 3, 3, True, True, 64, 112, 112),
    "ccfT_features3_24_40_k5_56": (24, 40, 3, 3, 5, True, True, 64, 56, 56),
    "ccfT_features4_40_80_k5_28": (40, 80, 6, 3, 5, True, True, 128, 28, 28),
    "ccfT_features5_80_96_k3_14": (80, 96, 6, 2, 3, False, True, 128, 14, 14),
    "ccfT_features6_96_192_k5_14": (96, 192, 6, 4, 5, True, True, 256, 14, 14),
    "ccfT_features7_192_320_k3_7": (192, 320, 6, 1, 3, False, True, 256, 7, 7),
}


@pytest.mark.parametrize("name", sorted(FULL_STAGES))
def test_stage_full_size_vs_mirror(name):
    """Well-conditioned state (projection BatchNorm weights x0.1).  The mirror runs on the host cores (10-40 s each)."""
    m, prog, st, shp = _stage_setup(name, 0.1, FULL_STAGES[name])
    x0 = C.det_input(shp)
    x = x0.cuda().requires_grad_(True)
    y = m(x)
    cot = C.cotangent(tuple(y.shape))
    (y * cot.cuda()).sum().backward()
    y_, dx_ = y.detach().cpu(), x.grad.cpu()
    grads = {kk: p.grad.cpu() for kk, p in m.named_parameters()}
    del y, x
    torch.cuda.empty_cache()
    r = M.run(prog, st, x0, True, cot, need_dx=True)
    ey, edx = rl2(y_, r["y"]), rl2(dx_, r["dx"])
    worst = max(rl2(gv, r["grads"][kk]) for kk, gv in grads.items() if not kk.endswith("conv.bias"))
    print(name, "y %.4f dx %.4f worst grad %.4f" % (ey, edx, worst))
    # four applications of the shared block (features6) accumulate a little more bf16 rounding than three: measured 0.030
    tol = 5e-2 if FULL_STAGES[name][3] >= 4 else TIGHT_BLK
    assert ey < 1e-2 and edx < tol
    for kk, gv in grads.items():
        if kk.endswith("conv.bias"):
            assert float(gv.abs().max()) < 1e-3
            continue
        e = rl2(gv, r["grads"][kk])
        assert e < (0.1 if kk.endswith("bn.weight") else tol), (kk, e)
    # ---- scale of every gradient tensor (round 6): the projection coefficient <g_hip, g_mirror> / |g_mirror|^2.  At stage scope
    # the rounding noise is 3-5e-2 of a tensor's norm and nearly orthogonal to it, so the projection sits within STAGE_PROJ of 1
    # (measured worst: see STAGE_PROJ) and a 10 % scale error in ONE tensor -- e.g. a BatchNorm-weight gradient, which the 0.1
    # rel-L2 bound above can let through -- is seen.  The whole network cannot resolve that for a 16..120-element tensor (its
    # per-tensor noise is 0.15-0.29, test_net_full_size_vs_mirror); every stage at its full size can.
    agree = _grad_agreement(grads, r["grads"])
    wp = max(agree.items(), key=lambda kv: abs(kv[1]["proj"] - 1.0))
    print("   projection coefficient: worst %.4f (%s, %d elements)" % (wp[1]["proj"], wp[0], wp[1]["n"]))
    if not os.environ.get("MNAS_SCALE_PRINT_ONLY"):
        for kk, v in agree.items():
            assert abs(v["proj"] - 1.0) <= STAGE_PROJ, (kk, v)
        victim = [kk for kk in grads if kk.endswith("bn.weight")][1]
        hurt = dict(grads)
        hurt[victim] = grads[victim] * 1.1                    # a deliberately injected x1.1 on one bn.weight gradient ...
        bad = [kk for kk, v in _grad_agreement(hurt, r["grads"]).items() if abs(v["proj"] - 1.0) > STAGE_PROJ]
        assert bad == [victim], bad                           # ... fails the check, and nothing else does


STAGE_PROJ = 0.02      # |projection - 1| per tensor at stage scope; MEASURED worst over the 14 full-size stages: 0.0079 (round 6, MI355X)


@pytest.mark.parametrize("name", ["features2_16_24_k3_112", "features3_24_40_k5_56", "features4_40_80_k5_28"])
def test_stage_bs256_replication_property(name):
    """BASELINE batch (bs 256) at the real spatial size WITHOUT a bs-256 CPU mirror (features.2 at 112x112 would need ~25 GB of
    host memory): a size-independent property.  A batch made of 4 copies of a 64-image batch has exactly the batch statistics
    of the 64-image batch, so every copy's output and input gradient must equal the bs-64 result and every parameter gradient
    must be 4x the bs-64 one (cotangent replicated as well); the bs-64 run itself is held to the mirror by
    test_stage_full_size_vs_mirror.  Differences come only from the fp32 summation order of the statistics (other partial
    tables at M = 3.2 M pixels): measured <= 1e-3; bound 5e-3 (y), 2e-2 (gradients)."""
    cin, cout, t, layers, k, reduce, ccf, N, H, W = FULL_STAGES[name]
    m, _, _, shp = _stage_setup(name, 0.1, FULL_STAGES[name])
    x0 = C.det_input(shp).cuda()
    x = x0.clone().requires_grad_(True)
    y = m(x)
    cot = C.cotangent(tuple(y.shape)).cuda()
    (y * cot).sum().backward()
    y64, dx64 = y.detach().clone(), x.grad.clone()
    g64 = {kk: p.grad.clone() for kk, p in m.named_parameters()}
    m.zero_grad(set_to_none=True)
    x = x0.repeat(4, 1, 1, 1).requires_grad_(True)
    y = m(x)
    (y * cot.repeat(4, 1, 1, 1)).sum().backward()
    for r in range(4):
        assert rl2(y[r * N:(r + 1) * N].detach().cpu(), y64.cpu()) < 5e-3, ("y", r)
        assert rl2(x.grad[r * N:(r + 1) * N].cpu(), dx64.cpu()) < 2e-2, ("dx", r)
    worst = 0.0
    for kk, p in m.named_parameters():
        if kk.endswith("conv.bias"):
            continue
        e = rl2(p.grad.cpu(), 4.0 * g64[kk].cpu())
        worst = max(worst, e)
        assert e < (5e-2 if kk.endswith("bn.weight") else 2e-2), (kk, e)
    print(name, "bs256 replication: worst parameter-gradient deviation %.5f" % worst)


@pytest.mark.parametrize("name", ["features3_24_40_k5_56", "features6_96_192_k5_14", "features7_192_320_k3_7"])
def test_side_stream_path_is_bit_identical(name):
    """Engine.use_side_stream (weight-gradient kernels on a second HIP stream; the default until round 3, now opt-in) must not
    change a single bit: the same kernels run in the same per-stream order and every accumulation into a gradient slice stays on
    one stream.  Outputs, input gradient, every parameter gradient and the BatchNorm buffers are compared with torch.equal."""
    spec = FULL_STAGES[name][:7] + (32,) + FULL_STAGES[name][8:]
    outs = []
    for side in (False, True):
        m, _, _, shp = _stage_setup(name, 0.1, spec)
        m._engine().use_side_stream = side
        x = C.det_input(shp).cuda().requires_grad_(True)
        y = m(x)
        cot = C.cotangent(tuple(y.shape)).cuda()
        (y * cot).sum().backward()
        torch.cuda.synchronize()
        outs.append((y.detach().cpu(), x.grad.cpu(), {kk: p.grad.cpu() for kk, p in m.named_parameters()},
                     {kk: v.cpu() for kk, v in m.state_dict().items() if "running" in kk}))
    (y0, dx0, g0, b0), (y1, dx1, g1, b1) = outs
    assert torch.equal(y0, y1) and torch.equal(dx0, dx1)
    for kk in g0:
        assert torch.equal(g0[kk], g1[kk]), kk
    for kk in b0:
        assert torch.equal(b0[kk], b1[kk]), kk


@pytest.mark.parametrize("name", sorted(C.STAGES))
def test_stage_bit_reproducible(name):
    """No float atomics anywhere on the path: outputs, input gradients and every parameter gradient of repeated runs
    on the same inputs are bit-identical (BatchNorm partial sums, weight-gradient splits and the fused reductions all
    use fixed summation orders)."""
    m, _, _, shp = _stage_setup(name, 1.0)
    x0 = C.det_input(shp).cuda()
    cot = None
    snaps = []
    for _ in range(3):
        m.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        y = m(x)
        if cot is None:
            cot = C.cotangent(tuple(y.shape)).cuda()
        (y * cot).sum().backward()
        snaps.append([y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in m.parameters()])
    for other in snaps[1:]:
        for a, b in zip(snaps[0], other):
            assert torch.equal(a, b)


@pytest.mark.parametrize("fused_pw", [False, True])
@pytest.mark.parametrize("name", sorted(C.NETS))
def test_net(name, fused_pw):
    """fused_pw=True forces the fused 1x1 backward (mnas_pw_bwd: input gradient + weight gradient + reduce in one sweep)
    for every supported layer regardless of its pixel count; by default it only engages at >= 100k pixels (bench sizes)."""
    from mnasnet_pytorch_amd import Mnasnet
    g = load("nets")
    ccf, N, H, W, train, pg = C.NETS[name]
    if fused_pw and (not train or pg == 1.0):
        pytest.skip("the fused backward is compared on the well-conditioned training cases")
    m = Mnasnet(cut_channels_first=ccf)
    assert list(m.state_dict().keys()) == O.state_keys(ccf)
    m.load_state_dict(O.init_state(ccf, C.STATE_SEED, proj_gamma=pg))
    m = m.cuda().train(train)
    if fused_pw:
        m.features._engine().pw_fused_min_pixels = 0
    x0 = C.det_input((N, 3, H, W))
    x = x0.cuda()
    prog, _ = O.build_program(ccf)
    st = O.init_state(ccf, C.STATE_SEED, proj_gamma=pg)
    gold_tol = 8e-2 if pg != 1.0 else 0.7
    if not train:
        with torch.no_grad():
            y = m(x)
        r = M.run(prog, st, x0, False)
        assert rl2(y.cpu(), r["y"]) < 2e-2
        assert rl2(y.cpu(), g[name + "/y"]) < 0.1        # eval mode: running statistics, no batch-stat feedback
        return
    y = m(x)
    cot = C.cotangent(tuple(y.shape))
    (y * cot.cuda()).sum().backward()
    r = M.run(prog, st, x0, True, cot)
    e_m, e_g = rl2(y.detach().cpu(), r["y"]), rl2(y.detach().cpu(), g[name + "/y"])
    print(name, "y vs mirror", e_m, "vs fp32 golden", e_g)
    assert e_g < gold_tol
    if pg == 1.0:
        return              # default state: gain ~110, only the sanity bound above is meaningful
    assert e_m < 3e-2
    coss = []
    for kk, p in m.named_parameters():
        if kk.endswith("conv.bias"):
            assert p.grad is not None
            continue
        a, b = p.grad.double().flatten().cpu(), r["grads"][kk].double().flatten()
        coss.append(float((a @ b) / (a.norm() * b.norm() + 1e-30)))
        assert 0.5 < float(a.norm() / b.norm()) < 2.0, kk
    print(name, "grad cosine vs mirror: min %.4f median %.4f" % (min(coss), float(np.median(coss))))
    assert min(coss) > 0.8 and np.median(coss) > 0.95
    cg = []
    for kk, p in m.named_parameters():
        if (name + "/g/" + kk) in g.files and not kk.endswith("conv.bias"):
            a, b = p.grad.double().flatten().cpu(), torch.as_tensor(g[name + "/g/" + kk]).double().flatten()
            cg.append(float((a @ b) / (a.norm() * b.norm() + 1e-30)))
    print(name, "grad cosine vs fp32 golden: min %.4f median %.4f" % (min(cg), float(np.median(cg))))
    assert np.median(cg) > 0.9
    sd = m.state_dict()
    for kk in g.files:
        if kk.startswith(name + "/") and "tracked" in kk:
            assert int(sd[kk[len(name) + 1:]]) == int(g[kk])
        if kk.startswith(name + "/ssum/") and pg != 1.0:
            v = sd[kk[len(name) + 6:]].double()
            assert abs(float(v.abs().sum()) - g[kk][1]) <= 5e-2 * g[kk][1], kk


def test_net_full_size_bit_reproducible():
    """Whole network at the BASELINE configuration (bs 256, 224x224): three forward+backward passes on the same inputs give
    bit-identical outputs and parameter gradients.  Any race between workgroups (LDS-DMA publication, cross-stream
    ordering, partial-table reuse) shows up here as a difference; small shapes do not expose them."""
    from mnasnet_pytorch_amd import Mnasnet
    m = Mnasnet(cut_channels_first=False)
    m.load_state_dict(O.init_state(False, C.STATE_SEED, proj_gamma=0.1))
    m = m.cuda().train()
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(256, 3, 224, 224, device="cuda", generator=g)
    snaps = []
    for _ in range(3):
        m.zero_grad(set_to_none=True)
        y = m(x)
        y.square().mean().backward()
        torch.cuda.synchronize()
        snaps.append([y.detach().clone()] + [p.grad.clone() for p in m.parameters()])
    names = ["y"] + [kk for kk, _ in m.named_parameters()]
    for other in snaps[1:]:
        for nm, a, b in zip(names, snaps[0], other):
            assert torch.equal(a, b), nm
            assert torch.isfinite(a).all(), nm


# Per-parameter relative L2 of the whole-network gradients against the mirror at 224x224, batch 32 (well-conditioned state):
# MEASURED on MI355X (round 5, printed by the test): ccf=False per-stage medians 0.17-0.24, worst tensor 0.288 (a bn.weight) /
# 0.273 (everything else); ccf=True medians 0.13-0.19, worst 0.280 / 0.267.  (The whole-network backward amplifies the 1-ulp bf16
# rounding-order differences between the engine and the mirror ~100x; per stage the same quantities are held to 3-5e-2 by
# test_stage_full_size_vs_mirror; a larger batch does not change it: MNAS_TEST_NET_N=96 measures worst 0.281 / 0.280.)  The bound is PER PARAMETER TENSOR, 1.5x the worst measurement: a wrong sign (rel-L2 2), a
# wrong scale (>= 0.5) or a dropped application of a shared block in ONE small tensor fails it -- a cosine median does not see that.
NET_GRAD_RL2 = {False: (0.42, 0.45), True: (0.42, 0.45)}      # {ccf: (bound for all, bound for bn.weight)}
# Scale agreement per tensor at WHOLE-NETWORK scope (round 6): bounds on |projection coefficient - 1| and
# |norm ratio - 1|, {suffix: (projection, norm ratio)} ~ 1.5x the MEASURED worst (MI355X, batch 32, both topologies):
#   conv.weight (432 .. 614 400 elements): projection 0.974 .. 0.996, norm ratio 0.991 .. 1.009
#   bn.bias     (16 .. 1920 elements):     projection 0.929 .. 1.048, norm ratio 0.950 .. 1.066
#   bn.weight   (16 .. 1920 elements):     projection 0.909 .. 1.089, norm ratio 0.934 .. 1.115
# -- the old bound was 0.5 .. 2.0.  The noise of the whole-network backward (rel-L2 0.15-0.29 per tensor) is nearly orthogonal to
# a big tensor's gradient, so a 10 % scale error in any conv weight gradient is seen here; in a 16..120-element BatchNorm tensor it
# moves the projection by up to 0.09 by itself (averaging over 6 inputs still leaves 0.056: tools/probe/scale_noise_k.py), so a
# 10 % error in ONE such tensor is resolved where the noise is small -- per stage at full size, STAGE_PROJ = 0.02 above.  What the
# network adds to the stages is the wiring between them: an error there scales EVERY gradient upstream of it, which the
# per-stage MEDIAN projection (noise averaged over 15-40 tensors) resolves to NET_STAGE_MEDIAN.
NET_GRAD_SCALE = {"bn.weight": (0.14, 0.17), "bn.bias": (0.11, 0.10), "conv.weight": (0.04, 0.02)}
NET_STAGE_MEDIAN = 0.05     # |median projection of a stage's tensors - 1|; MEASURED: 0.968 .. 0.991 (ccf=False), 0.981 .. 1.015 (ccf=True)


def _grad_agreement(grads, ref):
    """Per parameter tensor (conv.bias excluded: exactly 0 under train-mode BN): rel-L2, cosine, projection coefficient
    <g, ref> / |ref|^2 and norm ratio |g| / |ref| against the mirror's gradient."""
    out = {}
    for kk, gv in grads.items():
        if kk.endswith("conv.bias"):
            continue
        a, b = gv.double().flatten(), ref[kk].double().flatten()
        nb2 = float(b @ b) + 1e-60
        out[kk] = {"rel": float((a - b).norm() / (b.norm() + 1e-30)), "cos": float((a @ b) / (a.norm() * b.norm() + 1e-30)),
                   "proj": float(a @ b) / nb2, "ratio": float(a.norm() / (b.norm() + 1e-30)), "n": a.numel()}
    return out


def _stage_medians(agree):
    """{stage: median projection coefficient over the stage's parameter tensors}."""
    by = {}
    for kk, v in agree.items():
        by.setdefault(kk.split(".")[1], []).append(v["proj"])
    return {kk: float(np.median(v)) for kk, v in by.items()}


def _scale_violations(agree):
    """[(tensor, projection, norm ratio)] outside NET_GRAD_SCALE."""
    bad = []
    for kk, v in agree.items():
        suffix = next(sfx for sfx in NET_GRAD_SCALE if kk.endswith(sfx))
        bp, br = NET_GRAD_SCALE[suffix]
        if abs(v["proj"] - 1.0) > bp or abs(v["ratio"] - 1.0) > br:
            bad.append((kk, round(v["proj"], 4), round(v["ratio"], 4)))
    return bad


@pytest.mark.parametrize("ccf", [False, True])
def test_net_full_size_vs_mirror(ccf):
    """Whole network (both topologies, well-conditioned state) at 224x224 with batch 32 against the bf16 mirror on the host cores
    (about a minute): the stem, all six stages at their real spatial sizes, the 7x7 stage and the stage-to-stage wiring."""
    from mnasnet_pytorch_amd import Mnasnet
    m = Mnasnet(cut_channels_first=ccf)
    m.load_state_dict(O.init_state(ccf, C.STATE_SEED, proj_gamma=0.1))
    m = m.cuda().train()
    x0 = C.det_input((int(os.environ.get("MNAS_TEST_NET_N", "32")), 3, 224, 224))
    y = m(x0.cuda())
    cot = C.cotangent(tuple(y.shape))
    (y * cot.cuda()).sum().backward()
    y_ = y.detach().cpu()
    grads = {kk: p.grad.cpu() for kk, p in m.named_parameters()}
    del y
    torch.cuda.empty_cache()
    prog, _ = O.build_program(ccf)
    st = O.init_state(ccf, C.STATE_SEED, proj_gamma=0.1)
    r = M.run(prog, st, x0, True, cot)
    e_y = rl2(y_, r["y"])
    agree = _grad_agreement(grads, r["grads"])
    coss = [v["cos"] for v in agree.values()]
    rels = {kk: v["rel"] for kk, v in agree.items()}
    # ---- scale check (round 6): the projection coefficient <g_hip, g_mirror> / |g_mirror|^2 and the norm ratio of EVERY tensor.
    # The rounding-order noise (rel-L2 ~0.27) is close to orthogonal to the gradient, so it barely moves the projection: a 10 %
    # scale error in ONE tensor -- which the rel-L2 bound (0.42) and the old 0.5..2.0 norm-ratio bound let through -- fails here.
    bad = _scale_violations(agree)
    wp = max(agree.items(), key=lambda kv: abs(kv[1]["proj"] - 1.0))
    wr = max(agree.items(), key=lambda kv: abs(kv[1]["ratio"] - 1.0))
    print("  projection coefficient: worst %.4f (%s, %d elements); norm ratio: worst %.4f (%s)"
          % (wp[1]["proj"], wp[0], wp[1]["n"], wr[1]["ratio"], wr[0]))
    for suffix in ("bn.weight", "bn.bias", "conv.weight"):
        sel = [v for kk, v in agree.items() if kk.endswith(suffix)]
        print("    %-11s projection in [%.4f, %.4f], norm ratio in [%.4f, %.4f]"
              % (suffix, min(v["proj"] for v in sel), max(v["proj"] for v in sel), min(v["ratio"] for v in sel), max(v["ratio"] for v in sel)))
    assert not bad, bad
    med = _stage_medians(agree)
    print("    per-stage median projection: " + "  ".join("features.%s %.4f" % (kk, v) for kk, v in sorted(med.items())))
    if not os.environ.get("MNAS_SCALE_PRINT_ONLY"):
        assert all(abs(v - 1.0) <= NET_STAGE_MEDIAN for v in med.values()), med
        # the checks must SEE what they are for.  (a) x1.1 on ONE conv weight gradient: flagged per tensor, nothing else is.
        victim = [kk for kk in grads if kk.startswith("features.6") and kk.endswith("sequence.1.conv.weight")][0]
        hurt = dict(grads)
        hurt[victim] = grads[victim] * 1.1
        bad_inj = _scale_violations(_grad_agreement(hurt, r["grads"]))
        assert [b[0] for b in bad_inj] == [victim], bad_inj
        # (b) a wiring error: x1.1 on the gradient handed from features.5 to features.4 scales everything upstream of it
        hurt = {kk: (gv * 1.1 if int(kk.split(".")[1]) <= 4 else gv) for kk, gv in grads.items()}
        med_inj = _stage_medians(_grad_agreement(hurt, r["grads"]))
        assert sorted(kk for kk, v in med_inj.items() if abs(v - 1.0) > NET_STAGE_MEDIAN) == ["0", "1", "2", "3", "4"], med_inj
    print("full-size net ccf=%s: y vs mirror %.4f, grad cosine min %.4f median %.4f" % (ccf, e_y, min(coss), float(np.median(coss))))
    by_stage = {}
    for kk, e in rels.items():
        by_stage.setdefault(kk.split(".")[1], []).append((e, kk))
    for stg in sorted(by_stage):
        es = sorted(by_stage[stg])
        print("  features.%s: per-parameter rel-L2 median %.4f max %.4f (%s)" % (stg, es[len(es) // 2][0], es[-1][0], es[-1][1]))
    worst_bn = max(e for kk, e in rels.items() if kk.endswith("bn.weight"))
    worst_other = max(e for kk, e in rels.items() if not kk.endswith("bn.weight"))
    print("  worst rel-L2: bn.weight %.4f, everything else %.4f" % (worst_bn, worst_other))
    assert e_y < 3e-2
    assert min(coss) > 0.8 and np.median(coss) > 0.95
    if NET_GRAD_RL2[ccf] is not None:
        b_all, b_bnw = NET_GRAD_RL2[ccf]
        for kk, e in rels.items():
            assert e < (b_bnw if kk.endswith("bn.weight") else b_all), (kk, e)
    # BatchNorm bookkeeping of the step: running statistics and the update counters of every layer
    sd = m.state_dict()
    worst = 0.0
    for kk, v in sd.items():
        if kk.endswith("num_batches_tracked"):
            assert int(v) == int(st[kk]), kk
        elif "running_" in kk:
            worst = max(worst, rl2(v.cpu(), st[kk]))
    print("full-size net: running statistics vs mirror, worst relative L2 %.4f" % worst)
    assert worst < 3e-2


def test_no_cpu_fallback():
    from mnasnet_pytorch_amd import Mnasnet
    m = Mnasnet()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 32, 32))


def test_grad_accumulation_semantics():
    """.grad behaves like autograd's: None -> set, existing -> accumulated, zero_grad() honoured."""
    from mnasnet_pytorch_amd import MBConv_block
    m = MBConv_block(16, 3, 3)
    fill(m, "block_16_3_3")
    m = m.cuda().train()
    x = C.det_input((2, 16, 14, 14)).cuda()
    m(x).sum().backward()
    g1 = [p.grad.clone() for p in m.parameters()]
    m(x).sum().backward()                       # second backward without zero_grad: accumulates
    for p, a in zip(m.parameters(), g1):
        if a.abs().max() > 0:
            assert rl2(p.grad.cpu(), (2 * a).cpu()) < 5e-2
    m.zero_grad(set_to_none=True)
    m(x).sum().backward()
    for p, a in zip(m.parameters(), g1):
        if a.abs().max() > 0:
            assert rl2(p.grad.cpu(), a.cpu()) < 5e-2
    # foreign .grad tensors are added into, like AccumulateGrad does
    m.zero_grad(set_to_none=True)
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    m(x).sum().backward()
    for p, a in zip(m.parameters(), g1):
        if a.abs().max() > 0:
            assert rl2((p.grad - 1).cpu(), a.cpu()) < 5e-2


@pytest.mark.parametrize("hw", [(192, 256), (256, 192), (256, 256)])
def test_net_rectangular_clusters_vs_mirror(hw):
    """BASELINE config 5: the resolution clusters of ImnetDataset (datasets.py:331-335: 384x512, 512x512, 512x384), here at
    size_ratio 0.5 (train.py --size_ratio 0.5: 192x256, 256x256, 256x192) so that the CPU mirror finishes in seconds; the
    kernels take H, W at run time and a different shape is just another compiled program.  Same tolerances as test_net on
    the well-conditioned state: output <= 3e-2 relative L2 vs the bf16 mirror, gradient cosine min > 0.8 / median > 0.95."""
    from mnasnet_pytorch_amd import Mnasnet
    H, W = hw
    ccf, N, pg = False, 4, 0.1
    m = Mnasnet(cut_channels_first=ccf)
    m.load_state_dict(O.init_state(ccf, C.STATE_SEED, proj_gamma=pg))
    m = m.cuda().train()
    x0 = C.det_input((N, 3, H, W))
    y = m(x0.cuda())
    assert tuple(y.shape) == (N, 320, H // 32, W // 32)
    cot = C.cotangent(tuple(y.shape))
    (y * cot.cuda()).sum().backward()
    prog, _ = O.build_program(ccf)
    r = M.run(prog, O.init_state(ccf, C.STATE_SEED, proj_gamma=pg), x0, True, cot)
    assert rl2(y.detach().cpu(), r["y"]) < 3e-2
    coss = []
    for kk, p in m.named_parameters():
        if kk.endswith("conv.bias"):
            continue
        a, b = p.grad.double().flatten().cpu(), r["grads"][kk].double().flatten()
        coss.append(float((a @ b) / (a.norm() * b.norm() + 1e-30)))
    print(hw, "grad cosine vs mirror: min %.4f median %.4f" % (min(coss), float(np.median(coss))))
    assert min(coss) > 0.8 and np.median(coss) > 0.95
    # the same module then runs another cluster's shape (mixed-shape batches): a second program, same parameters
    y2 = m(C.det_input((2, 3, W, H)).cuda())
    assert tuple(y2.shape) == (2, 320, W // 32, H // 32) and bool(torch.isfinite(y2).all())


@pytest.mark.parametrize("hw", [(384, 512), (512, 512), (512, 384)])
def test_net_rectangular_clusters_full_size(hw):
    """BASELINE config 5 at the clusters' REAL sizes (datasets.py:331-335), batch 2, against the bf16 mirror (the mirror's
    cost at 512x512x2 is that of ~10 images at 224x224), same tolerances as the half-size case; plus the properties that do
    not need an oracle: finite everything, BatchNorm counters advanced once per application."""
    from mnasnet_pytorch_amd import Mnasnet
    H, W = hw
    ccf, N, pg = False, 2, 0.1
    m = Mnasnet(cut_channels_first=ccf)
    m.load_state_dict(O.init_state(ccf, C.STATE_SEED, proj_gamma=pg))
    m = m.cuda().train()
    x0 = C.det_input((N, 3, H, W))
    y = m(x0.cuda())
    assert tuple(y.shape) == (N, 320, H // 32, W // 32)
    cot = C.cotangent(tuple(y.shape))
    (y * cot.cuda()).sum().backward()
    y_ = y.detach().cpu()
    grads = {kk: p.grad.cpu() for kk, p in m.named_parameters()}
    assert all(bool(torch.isfinite(v).all()) for v in grads.values())
    prog, _ = O.build_program(ccf)
    st = O.init_state(ccf, C.STATE_SEED, proj_gamma=pg)
    r = M.run(prog, st, x0, True, cot)
    e_y = rl2(y_, r["y"])
    coss = []
    for kk, gv in grads.items():
        if kk.endswith("conv.bias"):
            continue
        a, b = gv.double().flatten(), r["grads"][kk].double().flatten()
        coss.append(float((a @ b) / (a.norm() * b.norm() + 1e-30)))
    print(hw, "full-size cluster: y %.4f, grad cosine min %.4f median %.4f" % (e_y, min(coss), float(np.median(coss))))
    assert e_y < 3e-2
    assert min(coss) > 0.8 and np.median(coss) > 0.95
    for kk, v in m.state_dict().items():
        if kk.endswith("num_batches_tracked"):
            assert int(v) == int(st[kk]), kk


K5_STAGES = {
    # BASELINE config 4, its 5x5 half: the stages that are 3x3 in MNASNet-1.0, with 5x5 depthwise convs
    "k5_features2_16_24": (16, 24, 3, 3, 5, True, False, 8, 112, 112),
    "k5_features5_80_96": (80, 96, 6, 2, 5, False, False, 16, 14, 14),
    "k5_features7_192_320": (192, 320, 6, 1, 5, False, False, 16, 7, 7),
}


@pytest.mark.parametrize("name", sorted(K5_STAGES))
def test_all_5x5_variant_stage_vs_mirror(name):
    """kernel_size=5 is a constructor argument of the reference's MBConv (mnasnet.py:139-147); the SE block of BASELINE config
    4 does not exist in the reference and is not built.  Well-conditioned state, tolerances of the stage tests."""
    m, prog, st, shp = _stage_setup(name, 0.1, K5_STAGES[name])
    x0 = C.det_input(shp)
    x = x0.cuda().requires_grad_(True)
    y = m(x)
    cot = C.cotangent(tuple(y.shape))
    (y * cot.cuda()).sum().backward()
    r = M.run(prog, st, x0, True, cot, need_dx=True)
    assert rl2(y.detach().cpu(), r["y"]) < 1e-2
    assert rl2(x.grad.cpu(), r["dx"]) < 5e-2
    for kk, p in m.named_parameters():
        if kk.endswith("conv.bias"):
            continue
        assert rl2(p.grad.cpu(), r["grads"][kk]) < (0.1 if kk.endswith("bn.weight") else 5e-2), kk


@pytest.mark.parametrize("hw,n", [((70, 58), 3), ((33, 95), 2), ((224, 224), 1)])
def test_net_odd_sizes_and_batch_one(hw, n):
    """Shapes no tuned path was sized for: odd / non-multiple-of-32 inputs (every stride-2 conv sees odd extents, so the
    parity-class input gradient falls back to the generic form, depthwise strips end in partial 4-column groups) and batch 1
    (BatchNorm over a single image).  Output vs the bf16 mirror <= 3e-2, gradient cosine median > 0.9."""
    from mnasnet_pytorch_amd import Mnasnet
    H, W = hw
    ccf, pg = False, 0.1
    m = Mnasnet(cut_channels_first=ccf)
    m.load_state_dict(O.init_state(ccf, C.STATE_SEED, proj_gamma=pg))
    m = m.cuda().train()
    x0 = C.det_input((n, 3, H, W))
    y = m(x0.cuda())
    cot = C.cotangent(tuple(y.shape))
    (y * cot.cuda()).sum().backward()
    prog, _ = O.build_program(ccf)
    r = M.run(prog, O.init_state(ccf, C.STATE_SEED, proj_gamma=pg), x0, True, cot)
    assert tuple(y.shape) == tuple(r["y"].shape)
    assert rl2(y.detach().cpu(), r["y"]) < 3e-2
    coss = []
    for kk, p in m.named_parameters():
        if kk.endswith("conv.bias"):
            continue
        a, b = p.grad.double().flatten().cpu(), r["grads"][kk].double().flatten()
        coss.append(float((a @ b) / (a.norm() * b.norm() + 1e-30)))
    print(hw, n, "grad cosine vs mirror: min %.4f median %.4f" % (min(coss), float(np.median(coss))))
    assert np.median(coss) > 0.9
