"""-m gpu: the native classifier head + loss (csrc/mnas_head.hip, mnasnet_pytorch_amd/head.py) against the CPU oracle
(oracle.head_forward_masked / F.cross_entropy, fp32) -- classifiers.py:56-89,107-111 and train.py:277,434-439.
Tolerances: fp32 products over K <= 1000 in a different summation order than ATen's: 2e-5 relative to the tensor's
max; the dropout keep mask must equal the oracle's restatement of the hash bit for bit."""
import ctypes as Ct

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cases as C
from cases import O
from gpu_util import relerr
from mnasnet_pytorch_amd import _lib as L
from mnasnet_pytorch_amd.head import NativeHead, parse_sequential, _mix

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _lin(N, I, O_, relu, p, seed, x, w, b):
    a = L.MnasHeadLinear()
    a.N, a.I, a.O, a.relu, a.drop_p, a.seed = N, I, O_, int(relu), p, seed
    a.x, a.w, a.b = x.data_ptr(), w.data_ptr(), L.ptr(b)
    return a


def _keep(seed, N, I, p):
    if p == 0:
        return torch.ones(N, I, dtype=torch.bool)
    return torch.from_numpy(O.head_dropout_keep(seed, N * I, p).reshape(N, I))


@pytest.mark.parametrize("N,I,O_", [(256, 320, 512), (37, 320, 1000), (5, 512, 10), (64, 256, 1000), (130, 70, 65)])
@pytest.mark.parametrize("p", [0.0, 0.5, 0.2])
def test_linear_fwd_bwd(N, I, O_, p):
    lib = L.load()
    seed = 0x1234567890ABCDEF ^ (N * 7919 + I)
    x, w, b = O.det_uniform((N, I), 1), O.det_uniform((O_, I), 2) * 0.1, O.det_uniform((O_,), 3)
    dz = O.det_uniform((N, O_), 4)
    u_prev = O.det_uniform((N, I), 5)                         # "input of this layer" as the ReLU mask source
    keep = _keep(seed, N, I, p)
    sc = 1.0 / (1.0 - float(np.float32(p)))
    xd = (x * keep * sc).double()
    xc, wc, bc, dzc, uc = (t.cuda() for t in (x, w, b, dz, u_prev))
    # device mask == oracle restatement
    if p > 0:
        mk = torch.empty(N * I, dtype=torch.uint8, device="cuda")
        L.check(lib.mnas_head_dropout_mask(mk.data_ptr(), N * I, p, seed, L.cur_stream()))
        assert torch.equal(mk.cpu().bool().view(N, I), keep)
        assert abs(float(keep.float().mean()) - (1 - p)) < 0.03
    for relu in (False, True):
        y = torch.full((N, O_), float("nan"), device="cuda")
        a = _lin(N, I, O_, relu, p, seed, xc, wc, bc)
        a.y = y.data_ptr()
        L.check(lib.mnas_head_linear_fwd(Ct.byref(a), L.cur_stream()))
        ref = xd @ w.double().t() + b.double()
        if relu:
            ref = ref.clamp_min(0)
        assert relerr(y.cpu(), ref) < TOL
    # weight / bias gradient, overwrite and accumulate
    for acc in (0, 1):
        dw = torch.full((O_, I), 0.25, device="cuda") if acc else torch.full((O_, I), float("nan"), device="cuda")
        db = torch.full((O_,), -0.5, device="cuda") if acc else torch.full((O_,), float("nan"), device="cuda")
        a = _lin(N, I, O_, False, p, seed, xc, wc, bc)
        a.dz, a.dw, a.db, a.accumulate = dzc.data_ptr(), dw.data_ptr(), db.data_ptr(), acc
        L.check(lib.mnas_head_linear_bwd_w(Ct.byref(a), L.cur_stream()))
        assert relerr(dw.cpu(), dz.double().t() @ xd + (0.25 if acc else 0.0)) < TOL
        assert relerr(db.cpu(), dz.double().sum(0) + (-0.5 if acc else 0.0)) < TOL
    # input gradient with and without the ReLU mask of the layer in front
    for masked in (False, True):
        dx = torch.full((N, I), float("nan"), device="cuda")
        a = _lin(N, I, O_, False, p, seed, xc, wc, bc)
        a.dz, a.dx = dzc.data_ptr(), dx.data_ptr()
        a.relu_mask = uc.data_ptr() if masked else None
        L.check(lib.mnas_head_linear_bwd_x(Ct.byref(a), L.cur_stream()))
        ref = (dz.double() @ w.double()) * keep * sc
        if masked:
            ref = ref * (u_prev > 0)
        assert relerr(dx.cpu(), ref) < TOL


@pytest.mark.parametrize("N,Cn", [(256, 1000), (7, 10), (33, 257)])
def test_cross_entropy(N, Cn):
    lib = L.load()
    x = O.det_uniform((N, Cn), 11) * 4
    t = torch.from_numpy((np.arange(N) * 7 + 3) % Cn).long()
    for ignore in (False, True):
        tt = t.clone()
        if ignore:
            tt[::3] = -100
        xr = x.clone().double().requires_grad_(True)
        ref = F.cross_entropy(xr, tt)
        ref.backward()
        rows = torch.empty(N, device="cuda")
        loss = torch.empty((), device="cuda")
        dl = torch.full((N, Cn), float("nan"), device="cuda")
        bad = torch.zeros(1, dtype=torch.int32, device="cuda")
        xc, tc = x.cuda(), tt.cuda()
        L.check(lib.mnas_head_cross_entropy(xc.data_ptr(), tc.data_ptr(), N, Cn, -100, rows.data_ptr(),
                                            loss.data_ptr(), dl.data_ptr(), bad.data_ptr(), L.cur_stream()))
        assert abs(float(loss) - float(ref.detach())) < 2e-5 * max(1.0, abs(float(ref.detach())))
        assert relerr(dl.cpu(), xr.grad) < 1e-5
        assert int(bad) == 0
    # out-of-range target: flagged, loss poisoned (ATen asserts on the device)
    tt = t.clone(); tt[1] = Cn
    xc, tc = x.cuda(), tt.cuda()
    L.check(lib.mnas_head_cross_entropy(xc.data_ptr(), tc.data_ptr(), N, Cn, -100, rows.data_ptr(),
                                        loss.data_ptr(), 0, bad.data_ptr(), L.cur_stream()))
    assert int(bad) == 1 and torch.isnan(loss).item()


def _classifier(cfg, num_classes=10):
    from mnasnet_pytorch_amd import FineTuneModelPool, load_model
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        m = FineTuneModelPool(load_model("mnasnet"), "mnasnet", num_classes, cfg)
    hst = O.init_head_state(cfg, num_classes, C.STATE_SEED)
    m.load_state_dict(hst, strict=False)
    return m.classifier.cuda(), hst


@pytest.mark.parametrize("cfg", C.HEADS)
@pytest.mark.parametrize("train", [False, True])
def test_head_module_matches_oracle(cfg, train):
    """NativeHead.apply (autograd path) on every classifier config: logits, input gradient and parameter gradients against the
    oracle run with the SAME keep masks (train) or without dropout (eval)."""
    seq, hst = _classifier(cfg)
    seq.train(train)
    head = NativeHead.build(seq)
    assert head is not None and len(head.layers) == sum(1 for l in O.HEAD_CONFIGS[cfg] if l[0] == "lin")
    N = 19
    f = (O.det_uniform((N, 320), 31).abs() * 2)
    fg = f.clone().cuda().requires_grad_(True)
    out = head.apply(fg)
    gl = O.det_uniform(tuple(out.shape), 32)
    out.backward(gl.cuda())
    keeps = None
    if train:
        seeds = [_mix(head.seed0, head.calls, i) for i in range(len(head.layers))]
        keeps = [_keep(s, N, l.lin.in_features, l.p) for s, l in zip(seeds, head.layers)]
        for i, l in enumerate(head.layers):      # the debug export agrees with what the GEMMs used
            assert torch.equal(head.dropout_mask(i, N, seeds[i]).cpu().bool(), keeps[i])
    fr = f.clone().double().requires_grad_(True)
    hd = {k: v.clone().double().requires_grad_(True) for k, v in hst.items()}
    ref = O.head_forward_masked(fr, hd, cfg, keeps)
    ref.backward(gl.double())
    assert relerr(out.detach().cpu(), ref.detach()) < TOL
    assert relerr(fg.grad.cpu(), fr.grad) < TOL
    for name, p in seq.named_parameters():
        assert relerr(p.grad.cpu(), hd["classifier." + name].grad) < TOL, name
    seq.zero_grad()


def test_native_step_matches_module_path():
    """Trainer.step without autograd (pool -> head -> cross-entropy -> backward as launch lists) against the same model
    stepped through model(x) / criterion / loss.backward(): identical kernels for the features, so loss, logits and every
    updated parameter agree to fp32 round-off of the head's gradient path."""
    from mnasnet_pytorch_amd.train_step import Trainer
    from test_gpu_train import build, _no_dropout
    x = C.det_input((4, 3, 64, 64)).cuda()
    t = torch.tensor([1, 3, 5, 7]).cuda()
    res = []
    for native in (True, False):
        m = build("512", proj_gamma=0.1).train(); _no_dropout(m)
        tr = Trainer(m, lr=1e-3)
        tr.native_step = native
        losses = [float(tr.step(x, t)) for _ in range(2)]
        res.append((losses, tr.flat_p.clone(), tr.last_logits))
        assert (tr._native_head() is not None) == native
    assert res[0][2] is not None and res[1][2] is None
    assert np.allclose(res[0][0], res[1][0], rtol=1e-5, atol=1e-6)
    assert relerr(res[0][1].cpu(), res[1][1].cpu()) < 1e-5


def test_native_step_dropout_runs_and_is_seeded():
    """with the reference's Dropout(0.5) active the step is reproducible under torch.manual_seed and differs across seeds"""
    from mnasnet_pytorch_amd.train_step import Trainer
    from test_gpu_train import build
    x = C.det_input((4, 3, 64, 64)).cuda()
    t = torch.tensor([1, 3, 5, 7]).cuda()
    out = []
    for seed in (5, 5, 6):
        torch.manual_seed(seed)
        m = build("512", proj_gamma=0.1).train()
        tr = Trainer(m, lr=1e-3)
        out.append([float(tr.step(x, t)) for _ in range(2)])
        assert all(np.isfinite(out[-1]))
    assert out[0] == out[1] and out[0] != out[2]
