"""Pins the CPU oracle (oracle/mnasnet_oracle.py) against the golden fixtures captured from the reference
(tests/golden/make_golden.py).  fp32; tolerance 1e-5 relative to the tensor's max magnitude -- the
restatement calls the same ATen ops, so at the generator's thread count it is bit-identical; the
tolerance only absorbs thread-count-dependent summation order."""
import numpy as np
import pytest
import torch

import cases as C
from cases import O

RTOL = 1e-5


@pytest.fixture(autouse=True)
def _one_thread():
    """The fixtures were generated with torch.set_num_threads(1); match it (results change bitwise with
    the thread count and tiny-batch BN amplifies that through 57 layers)."""
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


def close(a, b, rtol=RTOL):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(np.abs(b).max(), 1e-6)
    err = np.abs(a - b).max() / scale
    assert err <= rtol, "rel err %.3e" % err


def load(name):
    return np.load("%s/%s.npz" % (C.GOLDEN_DIR, name))


def prim_state(name, spec, seed=C.STATE_SEED):
    st = {}
    for suf, shp in (("conv.weight", spec.weight_shape()), ("conv.bias", (spec.cout,)), ("bn.weight", (spec.cout,)),
                     ("bn.bias", (spec.cout,)), ("bn.running_mean", (spec.cout,)), ("bn.running_var", (spec.cout,))):
        st[spec.prefix + "." + suf] = O.det_param(name + "." + suf, shp, seed)
    st[spec.prefix + ".bn.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    return st


def req_grad(st):
    for k, v in st.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)


@pytest.mark.parametrize("name", sorted(C.PRIMITIVES))
@pytest.mark.parametrize("train", [True, False])
def test_primitive(name, train):
    g = load("primitives")
    cin, cout, k, s, p, grp, N, H, W = C.PRIMITIVES[name]
    spec = O.ConvSpec("cb", cin, cout, k, s, p, grp)
    st = prim_state(name, spec)
    req_grad(st)
    x = C.det_input((N, cin, H, W)).requires_grad_(True)
    y = O.convblock(x, st, spec, train)
    (y * C.cotangent(tuple(y.shape))).sum().backward()
    tag = name + ("/train" if train else "/eval")
    close(y.detach(), g[tag + "/y"])
    close(x.grad, g[tag + "/dx"])
    for suf in ("conv.weight", "conv.bias", "bn.weight", "bn.bias"):
        ref = g[tag + "/d_" + suf]
        if suf == "conv.bias" and train:     # mathematically zero, numerically noise (SURVEY mismatch table)
            assert np.abs(st["cb." + suf].grad.numpy()).max() < 1e-4 and np.abs(ref).max() < 1e-4
            continue
        close(st["cb." + suf].grad, ref)
    close(st["cb.bn.running_mean"].detach(), g[tag + "/bn.running_mean"])
    close(st["cb.bn.running_var"].detach(), g[tag + "/bn.running_var"])
    assert int(st["cb.bn.num_batches_tracked"]) == int(g[tag + "/bn.num_batches_tracked"])


def block_state(name, specs):
    st = {}
    for j, s in enumerate(specs):
        for suf, shp in (("conv.weight", s.weight_shape()), ("conv.bias", (s.cout,)), ("bn.weight", (s.cout,)),
                         ("bn.bias", (s.cout,)), ("bn.running_mean", (s.cout,)), ("bn.running_var", (s.cout,))):
            st[s.prefix + "." + suf] = O.det_param("%s.sequence.%d.%s" % (name, j, suf), shp, C.STATE_SEED)
        st[s.prefix + ".bn.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    return st


@pytest.mark.parametrize("name", sorted(C.BLOCKS))
def test_block(name):
    g = load("blocks")
    c, t, k, N, H, W = C.BLOCKS[name]
    specs = O._block_specs("blk", c, t, k)
    st = block_state(name, specs)
    req_grad(st)
    x = C.det_input((N, c, H, W)).requires_grad_(True)
    y = O.mbconv_block(x, st, specs, True)
    (y * C.cotangent(tuple(y.shape))).sum().backward()
    close(y.detach(), g[name + "/y"])
    close(x.grad, g[name + "/dx"])
    for j in range(3):
        for suf in ("conv.weight", "bn.weight", "bn.bias"):
            close(st["blk.sequence.%d.%s" % (j, suf)].grad, g["%s/d_sequence.%d.%s" % (name, j, suf)])
        for suf in ("bn.running_mean", "bn.running_var"):
            close(st["blk.sequence.%d.%s" % (j, suf)].detach(), g["%s/sequence.%d.%s" % (name, j, suf)])


@pytest.mark.parametrize("name", sorted(C.SEPCONVS))
@pytest.mark.parametrize("train", [True, False])
def test_sepconv(name, train):
    """SepConv (mnasnet.py:64-103) on its own, incl. the list-multiplied pair (repeat >= 1): shared weights accumulate the
    gradients of their applications, shared BatchNorm buffers are updated once per application."""
    g = load("sepconvs")
    cin, cout, k, reduce, repeat, N, H, W = C.SEPCONVS[name]
    prog, uniq = C.sepconv_specs(name)
    st = C.sepconv_state(name, uniq)
    req_grad(st)
    x = C.det_input((N, cin, H, W)).requires_grad_(True)
    h = x
    for _, spec in prog:
        h = O.convblock(h, st, spec, train)
    (h * C.cotangent(tuple(h.shape))).sum().backward()
    tag = name + ("/train" if train else "/eval")
    close(h.detach(), g[tag + "/y"])
    close(x.grad, g[tag + "/dx"])
    for s_ in uniq:
        tail = s_.prefix[len("sep."):]
        for suf in ("conv.weight", "bn.weight", "bn.bias"):
            close(st[s_.prefix + "." + suf].grad, g["%s/d_%s.%s" % (tag, tail, suf)])
        for al in s_.aliases:                      # every alias of a shared ConvBlock shows the same buffers
            t2 = al[len("sep."):]
            close(st[s_.prefix + ".bn.running_mean"].detach(), g["%s/%s.bn.running_mean" % (tag, t2)])
            close(st[s_.prefix + ".bn.running_var"].detach(), g["%s/%s.bn.running_var" % (tag, t2)])
            assert int(st[s_.prefix + ".bn.num_batches_tracked"]) == int(g["%s/%s.bn.num_batches_tracked" % (tag, t2)])
    if train and repeat >= 2:
        assert int(st["sep.sequence.0.bn.num_batches_tracked"]) == repeat


def stage_forward(name, x, train=True):
    """Restates MBConv (mnasnet.py:139-173) for one stand-alone stage with the oracle's primitives."""
    cin, cout, t, layers, k, reduce, ccf, N, H, W = C.STAGES[name]
    stride = 2 if reduce else 1
    bc = cout if ccf else cin
    conv_idx = 0 if ccf else layers
    blk_idx = 1 if ccf else 0
    conv = O.ConvSpec("sequence.%d" % conv_idx, cin, cout, 3, stride, 1, 1)
    blk = O._block_specs("sequence.%d" % blk_idx, bc, t, k)
    st = {}
    for s in [conv] + blk:
        for suf, shp in (("conv.weight", s.weight_shape()), ("conv.bias", (s.cout,)), ("bn.weight", (s.cout,)),
                         ("bn.bias", (s.cout,)), ("bn.running_mean", (s.cout,)), ("bn.running_var", (s.cout,))):
            st[s.prefix + "." + suf] = O.det_param("%s.%s.%s" % (name, s.prefix, suf), shp, C.STATE_SEED)
        st[s.prefix + ".bn.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    req_grad(st)
    h = x
    if ccf:
        h = O.convblock(h, st, conv, train)
    for _ in range(layers):
        h = O.mbconv_block(h, st, blk, train)
    if not ccf:
        h = O.convblock(h, st, conv, train)
    return h, st


@pytest.mark.parametrize("name", sorted(C.STAGES))
def test_stage(name):
    g = load("stages")
    cin, cout, t, layers, k, reduce, ccf, N, H, W = C.STAGES[name]
    x = C.det_input((N, cin, H, W)).requires_grad_(True)
    y, st = stage_forward(name, x)
    (y * C.cotangent(tuple(y.shape))).sum().backward()
    close(y.detach(), g[name + "/y"])
    close(x.grad, g[name + "/dx"])
    n = 0
    for key in g.files:
        if key.startswith(name + "/d_") and not key.endswith("conv.bias"):
            close(st[key[len(name) + 3:]].grad, g[key])
            n += 1
        elif key.startswith(name + "/") and "running" in key and key[len(name) + 1:] in st:
            close(st[key[len(name) + 1:]].detach(), g[key])
            n += 1
        elif key.startswith(name + "/") and "tracked" in key and key[len(name) + 1:] in st:
            # shared block: updated `layers` times per forward
            assert int(st[key[len(name) + 1:]]) == int(g[key])
            if "sequence.%d.sequence" % (1 if ccf else 0) in key:
                assert int(g[key]) == layers
    assert n >= 16


@pytest.mark.parametrize("name", sorted(C.NETS))
def test_net(name):
    g = load("nets")
    ccf, N, H, W, train, pg = C.NETS[name]
    st = O.init_state(ccf, C.STATE_SEED, proj_gamma=pg)
    x = C.det_input((N, 3, H, W))
    if not train:
        with torch.no_grad():
            y = O.features_forward(x, st, ccf, False)
        close(y, g[name + "/y"])
        close(C.summarize(y), g[name + "/ysum"], 1e-4)
        return
    req_grad(st)
    y = O.features_forward(x, st, ccf, True)
    (y * C.cotangent(tuple(y.shape))).sum().backward()
    close(y.detach(), g[name + "/y"])
    checked = 0
    for key in g.files:
        if key.startswith(name + "/g/") and not key.endswith("conv.bias"):
            close(st[key[len(name) + 3:]].grad, g[key], 1e-4)
            checked += 1
        elif key.startswith(name + "/gsum/") and not key.endswith("conv.bias"):
            got = C.summarize(st[key[len(name) + 6:]].grad)
            ref = g[key]
            assert abs(got[2] - ref[2]) <= 1e-4 * max(ref[2], 1e-12)       # L2
            assert abs(got[1] - ref[1]) <= 1e-4 * max(ref[1], 1e-12)       # L1
            checked += 1
        elif key.startswith(name + "/ssum/"):
            got = C.summarize(st[key[len(name) + 6:]])
            assert abs(got[1] - g[key][1]) <= 1e-5 * max(g[key][1], 1e-12)
        elif "tracked" in key and key.startswith(name + "/"):
            assert int(st[key[len(name) + 1:]]) == int(g[key])
    assert checked > 90


def test_state_dict_keys():
    txt = open(C.GOLDEN_DIR + "/state_dict_keys.txt").read().split("# cut_channels_first=")
    for chunk in txt[1:]:
        lines = chunk.strip().split("\n")
        ccf = lines[0].strip() == "True"
        ref = [l.split() for l in lines[1:]]
        keys = O.state_keys(ccf)
        assert len(ref) == 399 and keys == [r[0] for r in ref]
        st = O.init_state(ccf)
        for k, shp, dt in ref:
            want = tuple(int(v) for v in shp.split("x")) if shp != "scalar" else ()
            assert tuple(st[k].shape) == want and str(st[k].dtype) == "torch." + dt
        # aliases share storage: 189 unique storages
        assert len({st[k].data_ptr() for k in keys if st[k].numel() > 0 and st[k].dim() > 0} |
                   {id(st[k]) for k in keys if st[k].dim() == 0}) == 189


@pytest.mark.parametrize("cfg", C.HEADS)
def test_head_eval(cfg):
    g = load("heads")
    net = O.OracleNet(ccf=False, head=cfg, num_classes=10, seed=C.STATE_SEED).eval()
    with torch.no_grad():
        y = net(C.det_input((2, 3, 64, 64)))
    close(y, g["head_%s/eval_logits" % cfg], 1e-4)


def test_train_step():
    """train.py:423-440 with Adam(lr=1e-3) + CrossEntropyLoss, two steps, dropout p=0."""
    g = load("heads")
    net = O.OracleNet(ccf=False, head="512", num_classes=10, seed=C.STATE_SEED).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    x = C.det_input((4, 3, 64, 64))
    target = torch.tensor([1, 3, 5, 7])
    crit = torch.nn.CrossEntropyLoss()
    losses = []
    for _ in range(2):
        out = net(x.float(), dropout=False)
        loss = crit(out, target)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    close(losses, g["step/losses"], 1e-4)
    st = {**net.state(), **net.head_state()}
    n = 0
    for key in g.files:
        if key.startswith("step/psum/"):
            got = C.summarize(st[key[len("step/psum/"):]])
            assert abs(got[1] - g[key][1]) <= 2e-4 * max(g[key][1], 1e-12), key
            n += 1
    assert n > 100
