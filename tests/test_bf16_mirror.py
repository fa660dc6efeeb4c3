"""CPU: pins oracle/bf16_mirror.py (explicit-backward, bf16-storage restatement) to the golden fixtures
captured from the reference, within the tolerance that bf16 storage legitimately costs (relative L2):
single ConvBlock <= 1e-2 outputs / 8e-2 gradients (tiny batches: 72..288 samples per channel); blocks <= 3e-2 / 0.2 (392 samples per channel: a single ReLU-mask flip moves a BN gradient by ~5 %).
This validates the hand-derived BatchNorm/ReLU/conv backward formulas the HIP kernels implement."""
import numpy as np
import pytest
import torch

import cases as C
from cases import O
from oracle import bf16_mirror as M
from test_oracle_golden import block_state, load, prim_state


def rl2(a, b):
    a = torch.as_tensor(np.asarray(a)).double().flatten()
    b = torch.as_tensor(np.asarray(b)).double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("name", sorted(C.PRIMITIVES))
def test_primitive(name):
    g = load("primitives")
    cin, cout, k, s, p, grp, N, H, W = C.PRIMITIVES[name]
    spec = O.ConvSpec("cb", cin, cout, k, s, p, grp)
    st = prim_state(name, spec)
    x = C.det_input((N, cin, H, W))
    for train in (True, False):
        st = prim_state(name, spec)
        tag = name + ("/train" if train else "/eval")
        yshape = g[tag + "/y"].shape
        r = M.run([("conv", spec)], st, x, train=train, cot=C.cotangent(tuple(yshape)) if train else None, need_dx=True)
        assert rl2(r["y"], g[tag + "/y"]) < 1e-2
        if not train:
            continue
        if cin != 3:
            assert rl2(r["dx"], g[tag + "/dx"]) < 1.2e-1
        assert rl2(r["grads"]["cb.conv.weight"], g[tag + "/d_conv.weight"]) < 1.2e-1
        assert rl2(r["grads"]["cb.bn.weight"], g[tag + "/d_bn.weight"]) < 1.2e-1
        assert rl2(r["grads"]["cb.bn.bias"], g[tag + "/d_bn.bias"]) < 1.2e-1
        assert rl2(st["cb.bn.running_mean"], g[tag + "/bn.running_mean"]) < 1e-2
        assert rl2(st["cb.bn.running_var"], g[tag + "/bn.running_var"]) < 1e-2


@pytest.mark.parametrize("name", sorted(C.BLOCKS))
def test_block(name):
    g = load("blocks")
    c, t, k, N, H, W = C.BLOCKS[name]
    specs = O._block_specs("blk", c, t, k)
    st = block_state(name, specs)
    x = C.det_input((N, c, H, W))
    r = M.run([("block", specs)], st, x, True, C.cotangent((N, c, H, W)), need_dx=True)
    assert rl2(r["y"], g[name + "/y"]) < 3e-2
    assert rl2(r["dx"], g[name + "/dx"]) < 1.2e-1
    for j in range(3):
        for suf in ("conv.weight", "bn.weight", "bn.bias"):
            assert rl2(r["grads"]["blk.sequence.%d.%s" % (j, suf)], g["%s/d_sequence.%d.%s" % (name, j, suf)]) < 0.2


@pytest.mark.parametrize("name", sorted(C.SEPCONVS))
def test_sepconv(name):
    g = load("sepconvs")
    cin, cout, k, reduce, repeat, N, H, W = C.SEPCONVS[name]
    prog, uniq = C.sepconv_specs(name)
    x = C.det_input((N, cin, H, W))
    for train in (True, False):
        st = C.sepconv_state(name, uniq)
        tag = name + ("/train" if train else "/eval")
        yshape = tuple(g[tag + "/y"].shape)
        r = M.run(prog, st, x, train=train, cot=C.cotangent(yshape) if train else None, need_dx=True)
        assert rl2(r["y"], g[tag + "/y"]) < 3e-2
        if not train:
            continue
        assert rl2(r["dx"], g[tag + "/dx"]) < 0.2
        for s_ in uniq:
            tail = s_.prefix[len("sep."):]
            for suf in ("conv.weight", "bn.weight", "bn.bias"):
                assert rl2(r["grads"][s_.prefix + "." + suf], g["%s/d_%s.%s" % (tag, tail, suf)]) < 0.2, (s_.prefix, suf)
            assert rl2(st[s_.prefix + ".bn.running_mean"], g["%s/%s.bn.running_mean" % (tag, tail)]) < 2e-2
            assert rl2(st[s_.prefix + ".bn.running_var"], g["%s/%s.bn.running_var" % (tag, tail)]) < 2e-2


@pytest.mark.parametrize("name", ["net_ccfF_b8_96_wc_train", "net_ccfT_b8_96_wc_train"])
def test_net_forward_and_grad_norms(name):
    g = load("nets")
    ccf, N, H, W, train, pg = C.NETS[name]
    st = O.init_state(ccf, C.STATE_SEED, proj_gamma=pg)
    prog, _ = O.build_program(ccf)
    x = C.det_input((N, 3, H, W))
    r = M.run(prog, st, x, True, C.cotangent(tuple(g[name + "/y"].shape)))
    ey = rl2(r["y"], g[name + "/y"])
    print(name, 'y rel-L2', ey)
    assert ey < 0.08      # 57 bf16 layers on the well-conditioned state (measured ~0.03)
    # shared blocks accumulated `layers` contributions; norms comparable with the fp32 reference
    n = 0
    for key in g.files:
        if key.startswith(name + "/gsum/") and not key.endswith("conv.bias"):
            kk = key[len(name) + 6:]
            got = float(r["grads"][kk].double().norm())
            ref = float(np.sqrt(g[key][2]))
            assert 0.4 * ref <= got <= 2.5 * ref, (kk, got, ref)
            n += 1
    assert n >= 81


def test_image_gradient_of_the_front():
    """need_dx on an image input (round 5: the HIP stem has an input gradient): the mirror's dL/d image of the network front (stem,
    SepConv, the 112x112 stage, the stride-2 conv) against fp32 autograd through the same layers.  bf16 storage puts the PARAMETER
    gradients of this pass at 0.10 (median) - 0.21 (max) of the fp32 ones; the image gradient sits among them (0.11-0.13): < 0.2.
    (The tight check is HIP vs this mirror: tests/test_gpu_train.py::test_image_gradient_vs_mirror.)"""
    import torch.nn.functional as F
    st = O.init_state(False, C.STATE_SEED, proj_gamma=0.1)
    prog, _ = O.build_program(False)
    front = prog[:7]
    x = C.det_input((2, 3, 40, 48))
    out = M.run(front, st, x, True)
    cot = C.cotangent(tuple(out["y"].shape))
    r = M.run(front, st, x, True, cot, need_dx=True)
    assert r["dx"] is not None and r["dx"].dtype == torch.float32 and tuple(r["dx"].shape) == tuple(x.shape)

    def cb(spec, a):                                      # fp32 ConvBlock, batch statistics: relu(bn(conv(x))) (mnasnet.py:58-62)
        p = spec.prefix
        y = F.conv2d(a, st[p + ".conv.weight"], st[p + ".conv.bias"], stride=spec.stride, padding=spec.pad, groups=spec.groups)
        y = F.batch_norm(y, None, None, st[p + ".bn.weight"], st[p + ".bn.bias"], training=True, eps=1e-5)
        return torch.relu(y)

    xr = x.clone().requires_grad_(True)
    cur = xr
    for op, arg in front:
        if op == "conv":
            cur = cb(arg, cur)
        else:
            h = cur
            for spec in arg[:3]:
                h = cb(spec, h)
            cur = cur + h
    (cur * cot).sum().backward()
    e = rl2(r["dx"], xr.grad)
    print("mirror image gradient vs fp32 autograd: rel-L2 %.4f" % e)
    assert e < 0.2
