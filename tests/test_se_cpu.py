"""CPU: the SE variant of BASELINE config 4 (5x5 depthwise + squeeze-excite).  The reference has NO SE block: the definition is
this repo's (mnasnet_pytorch_amd.mnasnet.SqueezeExcite), restated in oracle.se_apply -- "parity unpinned by the reference".
These tests pin the restatements to each other: module key set == oracle key set, and the bf16 mirror's explicit SE backward ==
autograd through the fp32 oracle within the bf16 tolerances of tests/test_bf16_mirror.py."""
import torch

import cases as C
from cases import O
from oracle import bf16_mirror as M


def rl2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_default_key_set_is_still_the_reference_one():
    from mnasnet_pytorch_amd import Mnasnet
    assert list(Mnasnet(False).state_dict().keys()) == O.state_keys(False)
    assert len(O.state_keys(False)) == 399


def test_se_variant_key_set_and_shapes():
    from mnasnet_pytorch_amd import Mnasnet
    m = Mnasnet(False, kernel_size=5, se_ratio=0.25)
    sd = m.state_dict()
    st = O.init_state(False, 1, kernel=5, se_ratio=0.25)
    assert set(sd.keys()) == set(st.keys())
    for k in sd:
        assert tuple(sd[k].shape) == tuple(st[k].shape), k
    m.load_state_dict(st)
    # shared (list-multiplied) blocks share their SE module too
    f2 = m.features[2].sequence
    assert f2[0].se is f2[1].se and f2[0].se.fc1.weight.shape == (8, 48)
    assert sum(1 for k in sd if ".se." in k) == 4 * 16       # 16 block applications x 4 tensors (aliases included)
    # every depthwise conv of the variant is 5x5
    assert all(mod.conv.kernel_size == (5, 5) for mod in m.modules()
               if type(mod).__name__ == "ConvBlock" and mod.conv.groups > 1 and mod.conv.in_channels > 32)


def test_mirror_se_backward_matches_autograd():
    """block with SE at 2 x 16 x 12 x 12: the mirror's hand-written backward (incl. se_bwd) against autograd through the fp32
    oracle on the same state: y <= 4e-2, dx and every parameter gradient <= 0.3 (bf16 storage, tiny batch), SE gradients <= 0.1."""
    c, t, k, N, H, W = 16, 3, 5, 2, 12, 12
    specs = O._block_specs("blk", c, t, k, se_ratio=0.25)
    assert len(specs) == 4 and specs[3].kind == "se" and specs[3].reduced == 8

    def state():
        st = {}
        for s in specs[:3]:
            for suf, shp in (("conv.weight", s.weight_shape()), ("conv.bias", (s.cout,)), ("bn.weight", (s.cout,)),
                             ("bn.bias", (s.cout,)), ("bn.running_mean", (s.cout,)), ("bn.running_var", (s.cout,))):
                st[s.prefix + "." + suf] = O.det_param("se." + s.prefix + "." + suf, shp, 3)
            st[s.prefix + ".bn.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
        se = specs[3]
        st[se.prefix + ".fc1.weight"] = O.det_param(se.prefix + ".fc1.weight", (se.reduced, se.channels), 3) * 4
        st[se.prefix + ".fc1.bias"] = O.det_param(se.prefix + ".fc1.bias", (se.reduced,), 3)
        st[se.prefix + ".fc2.weight"] = O.det_param(se.prefix + ".fc2.weight", (se.channels, se.reduced), 3) * 4
        st[se.prefix + ".fc2.bias"] = O.det_param(se.prefix + ".fc2.bias", (se.channels,), 3)
        return st
    x0 = C.det_input((N, c, H, W))
    st = state()
    r = M.run([("block", specs)], st, x0, True, None)
    cot = C.cotangent(tuple(r["y"].shape))
    st = state()
    r = M.run([("block", specs)], st, x0, True, cot, need_dx=True)
    st2 = state()
    for v in st2.values():
        if v.dtype.is_floating_point:
            v.requires_grad_(True)
    x = x0.clone().requires_grad_(True)
    with torch.no_grad():
        pass
    params = {k: v for k, v in st2.items() if v.dtype.is_floating_point and "running" not in k}
    st_run = {k: (v if k in params else v.detach().clone()) for k, v in st2.items()}
    y = O.mbconv_block(x, st_run, specs, True)
    (y * cot).sum().backward()
    assert rl2(r["y"], y.detach()) < 4e-2
    assert rl2(r["dx"], x.grad) < 0.3
    for k, v in params.items():
        if k.endswith("conv.bias"):
            continue
        tol = 0.1 if ".se." in k else 0.3
        assert rl2(r["grads"][k], v.grad) < tol, (k, rl2(r["grads"][k], v.grad))


def test_mirror_se_on_load_backward_matches_materialised():
    """The mirror's two restatements of the squeeze-excite block's backward -- the materialised a*s path and the excitation-on-load
    path (per-image weight-gradient slabs gated in fp32, du from the slabs; optionally dL/da2 left unrounded for the depthwise
    backward) -- are the same mathematics with different rounding points: identical forward, gradients within 5e-2 of each other
    (BatchNorm weight gradients, sums with heavy cancellation at this tiny size: 0.1 -- the bounds of tests/test_gpu_se.py)."""
    c, t, k, N, H, W = 16, 3, 5, 3, 10, 10
    specs = O._block_specs("blk", c, t, k, se_ratio=0.25)

    def state():
        st = {}
        for s in specs[:3]:
            for suf, shp in (("conv.weight", s.weight_shape()), ("conv.bias", (s.cout,)), ("bn.weight", (s.cout,)),
                             ("bn.bias", (s.cout,)), ("bn.running_mean", (s.cout,)), ("bn.running_var", (s.cout,))):
                st[s.prefix + "." + suf] = O.det_param("se2." + s.prefix + "." + suf, shp, 5)
            st[s.prefix + ".bn.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
        se = specs[3]
        st[se.prefix + ".fc1.weight"] = O.det_param(se.prefix + ".fc1.weight", (se.reduced, se.channels), 5) * 4
        st[se.prefix + ".fc1.bias"] = O.det_param(se.prefix + ".fc1.bias", (se.reduced,), 5)
        st[se.prefix + ".fc2.weight"] = O.det_param(se.prefix + ".fc2.weight", (se.channels, se.reduced), 5) * 4
        st[se.prefix + ".fc2.bias"] = O.det_param(se.prefix + ".fc2.bias", (se.channels,), 5)
        return st
    x0 = C.det_input((N, c, H, W))
    cot = C.cotangent((N, c, H, W))
    base = M.run([("block", specs)], state(), x0, True, cot, need_dx=True)
    r = M.run([("block", specs)], state(), x0, True, cot, need_dx=True, se_on_load=lambda *a: True)
    assert torch.equal(r["y"], base["y"])
    assert rl2(r["dx"], base["dx"]) < 5e-2
    for kk, v in base["grads"].items():
        if not kk.endswith("conv.bias"):
            assert rl2(r["grads"][kk], v) < (0.1 if kk.endswith("bn.weight") else 5e-2), kk


def test_se_segments_per_image():
    """Host logic of the excitation-on-load backward: workgroups per image for the segment-mode mnas_pw_bwd.  The choices at the
    bench shapes (bs 256; tile pixels / channel slices as csrc/mnas_pwbwd.hip reports them) and the constraints: a divisor of H*W,
    slab table within the scratch, 0 when nothing fits."""
    from mnasnet_pytorch_amd.engine import se_segments_per_image as seg
    assert seg(256, 112 * 112, 128, 1, 1 << 20) == 2          # 48 -> 16: 6272 px = 49 whole tiles, one round of 512 workgroups
    assert seg(256, 56 * 56, 128, 1, 1 << 20) == 2            # 72 -> 24: 1568 px = 12.25 tiles (6 % ragged), 512 workgroups
    assert seg(256, 28 * 28, 64, 3, 1 << 20) == 2             # 240 -> 40, 3 channel slices: 1536 workgroups = 3 whole rounds
    for N, HW, tile, sl in ((256, 12544, 128, 1), (64, 3136, 128, 1), (8, 784, 64, 3), (3, 196, 64, 6)):
        d = seg(N, HW, tile, sl, 1 << 20)
        assert d >= 1 and HW % d == 0 and N * d <= 4096
    assert seg(256, 784, 64, 3, 255) == 0                     # not even one slab per image fits
    assert seg(256, 784, 64, 3, 256) == 1


def test_gate_and_segment_queries_host_side():
    """The C ABI's host-side queries the engine sizes the on-load path with (no launch): gated-forward support of the project conv
    shapes, tile pixels / channel slices of the fused 1x1 backward."""
    from mnasnet_pytorch_amd import _lib as L
    lib = L.load()
    for H, Ci, Co in ((112, 48, 16), (56, 72, 24), (28, 240, 40)):
        assert lib.mnas_conv_gemm_gate_ok(256, H * H, Ci, Co) == 1
    assert lib.mnas_conv_gemm_gate_ok(256, 14 * 14, 576, 96) == 0 and lib.mnas_conv_gemm_gate_ok(256, 7 * 7, 1152, 192) == 0
    assert lib.mnas_conv_gemm_gate_ok(256, 12544, 20, 16) == 0          # channels not a multiple of 8
    assert (lib.mnas_pw_bwd_tile_pixels(48, 16), lib.mnas_pw_bwd_slices(48, 16)) == (128, 1)
    assert (lib.mnas_pw_bwd_tile_pixels(240, 40), lib.mnas_pw_bwd_slices(240, 40)) == (64, 3)
    assert (lib.mnas_pw_bwd_tile_pixels(576, 96), lib.mnas_pw_bwd_slices(576, 96)) == (64, 6)
    assert lib.mnas_pw_bwd_tile_pixels(96, 576) == -1
