"""-m gpu: FineTuneModelPool + the train step (train.py:423-440) on the HIP engine vs the goldens captured from
the reference (tests/golden/heads.npz) and vs the CPU oracle.  Tolerances: logits <= 6e-2 relative L2 (bf16
features through 57 layers feeding an fp32 head; default, ill-conditioned state), loss <= 3 %."""
import numpy as np
import pytest
import torch

import cases as C
from cases import O

pytestmark = pytest.mark.gpu


def rl2(a, b):
    a = torch.as_tensor(np.asarray(a)).double().flatten()
    b = torch.as_tensor(np.asarray(b)).double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def build(cfg, num_classes=10, proj_gamma=1.0):
    import contextlib, io
    from mnasnet_pytorch_amd import FineTuneModelPool, load_model
    with contextlib.redirect_stdout(io.StringIO()):
        base = load_model("mnasnet")
    m = FineTuneModelPool(base, "mnasnet", num_classes, cfg)
    m.load_state_dict({**O.init_state(False, C.STATE_SEED, proj_gamma=proj_gamma),
                       **O.init_head_state(cfg, num_classes, C.STATE_SEED)})
    return m.cuda()


@pytest.mark.parametrize("cfg", C.HEADS)
def test_head_eval_logits(cfg):
    g = np.load(C.GOLDEN_DIR + "/heads.npz")
    m = build(cfg).eval()
    with torch.no_grad():
        y = m(C.det_input((2, 3, 64, 64)).cuda())
    assert rl2(y.cpu(), g["head_%s/eval_logits" % cfg]) < 6e-2


def _no_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0


def test_train_step_with_torch_optimizer():
    """Autograd contract: loss.backward() fills .grad of the SAME Parameters a stock optimizer holds."""
    g = np.load(C.GOLDEN_DIR + "/heads.npz")
    m = build("512").train()
    _no_dropout(m)
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, m.parameters()), lr=1e-3)
    crit = torch.nn.CrossEntropyLoss()
    x = C.det_input((4, 3, 64, 64)).cuda()
    target = torch.tensor([1, 3, 5, 7]).cuda()
    losses = []
    for _ in range(2):
        out = m(x.float())
        loss = crit(out, target)
        opt.zero_grad()
        loss.backward()
        assert all(p.grad is not None for p in m.parameters())
        opt.step()
        losses.append(float(loss.detach()))
    ref = g["step/losses"]
    assert abs(losses[0] - ref[0]) <= 3e-2 * abs(ref[0]), (losses, ref)
    assert abs(losses[1] - ref[1]) <= 6e-2 * abs(ref[1]), (losses, ref)      # after one Adam step on a gain-110 net
    assert losses[1] < losses[0]


def test_trainer_matches_torch_optimizer_path():
    """Trainer (flat buffers + fused Adam) == the stock-optimizer path, step for step."""
    from mnasnet_pytorch_amd.train_step import Trainer
    x = C.det_input((4, 3, 64, 64)).cuda()
    target = torch.tensor([1, 3, 5, 7]).cuda()
    # well-conditioned state: the LDS float atomics in the statistics reductions make runs differ in the last
    # ulp, which the default (gain ~110) state amplifies to ~0.3 % of the loss at this tiny batch
    m1 = build("512", proj_gamma=0.1).train(); _no_dropout(m1)
    opt = torch.optim.Adam(m1.parameters(), lr=1e-3)
    crit = torch.nn.CrossEntropyLoss()
    l1 = []
    for _ in range(3):
        loss = crit(m1(x), target); opt.zero_grad(); loss.backward(); opt.step(); l1.append(float(loss.detach()))
    m2 = build("512", proj_gamma=0.1).train(); _no_dropout(m2)
    tr = Trainer(m2, lr=1e-3)
    l2 = [float(tr.step(x, target)) for _ in range(3)]
    assert abs(l1[0] - l2[0]) <= 5e-3 * abs(l1[0]), (l1, l2)
    assert abs(l1[2] - l2[2]) <= 3e-2 * abs(l1[2]), (l1, l2)
    # parameters are views of one flat buffer and still the module's own Parameter objects
    assert m2.features[0].conv.weight.data_ptr() >= tr.flat_p.data_ptr()
    sd = m2.state_dict()
    assert list(k for k in sd if k.startswith("features")) == O.state_keys(False)


def test_baseline_config_step_bs256():
    """BASELINE configs[1] at its real size: Trainer.step at bs 256 x 3 x 224 x 224 (head '512', 1000 classes, Adam).  The batch
    is 4 copies of a 64-image batch (dropout off): the mean-reduced loss and every gradient must equal the bs-64 step's
    (identical BatchNorm statistics; mean over 4x the rows of 4x repeated terms), which pins the M = 3.2 M-pixel persistent grids
    of the real bench shape to the bs-64 grids that the mirror tests check.  The only legitimate difference is the fp32 summation
    order of the statistics tables, which flips a few 1-ulp bf16 roundings that the 57-layer backward then amplifies (the same
    effect that makes the whole-network tests compare gradients by cosine): loss within 2e-3 relative (measured 2e-4), flat
    gradient cosine >= 0.99 and relative L2 <= 0.2 (measured 0.995 / 0.098); the per-stage replication tests in
    test_gpu_model.py hold the same property to 2e-2."""
    from mnasnet_pytorch_amd.train_step import Trainer
    g = torch.Generator(device="cuda").manual_seed(3)
    x64 = torch.randn(64, 3, 224, 224, device="cuda", generator=g)
    t64 = torch.randint(0, 1000, (64,), device="cuda", generator=g)
    out = []
    for rep in (1, 4):
        m = build("512", 1000, proj_gamma=0.1).train()
        _no_dropout(m)
        tr = Trainer(m, lr=1e-3)
        x, t = x64.repeat(rep, 1, 1, 1), t64.repeat(rep)
        loss = float(tr.step(x, t))
        torch.cuda.synchronize()
        out.append((loss, tr.flat_g.clone(), tr.flat_p.clone()))
        del tr, m
        torch.cuda.empty_cache()
    (l1, g1, p1), (l4, g4, p4) = out
    assert np.isfinite(l4) and abs(l4 - l1) <= 2e-3 * abs(l1), (l1, l4)
    a, b = g4.double().cpu(), g1.double().cpu()
    cos = float((a @ b) / (a.norm() * b.norm()))
    print("bs256 vs 4 x bs64: loss %.5f / %.5f, flat gradient cosine %.5f, relative L2 %.4f" % (l4, l1, cos, rl2(g4.cpu(), g1.cpu())))
    assert cos > 0.99 and rl2(g4.cpu(), g1.cpu()) < 0.2
    assert bool(torch.isfinite(p4).all())


@pytest.mark.parametrize("nb,bs,size,steps", [(4, 32, 64, 30), (2, 48, 224, 6)], ids=["bs32_64px_30steps", "bs48_224px_6steps"])
def test_training_trajectory_matches_cpu_oracle(nb, bs, size, steps):
    """What "matches the reference" means for bf16 TRAINING (train.py:423-440): Adam steps (lr 1e-3), 10 classes, dropout off,
    cycling over `nb` fixed batches, HIP Trainer vs the fp32 CPU oracle (oracle.train_step) from the same state on identical data
    -- 30 steps at bs 32, 64x64, and (round 4; 6 steps at bs 48: the CPU oracle is what this test's time goes to) the
    bench resolution, 224 x 224.  Stated band: at every step
    |loss_hip - loss_ref| <= max(6 % of loss_ref, 0.03) (bf16 activations, fp32 master weights / statistics / optimizer; the
    absolute floor covers the end of the long run, where the 128 images are memorised and the loss is ~0.01), the mean gap over
    the steps with loss_ref > 0.1 is <= 3 %, and both runs learn (long run: last-5 mean < 0.2 x first-5 mean; short run: the last
    loss is below the first).  Measured on MI355X: see the printed curve."""
    from mnasnet_pytorch_amd.train_step import Trainer
    torch.manual_seed(0)
    gen = torch.Generator().manual_seed(123)
    xs = [torch.randn(bs, 3, size, size, generator=gen) for _ in range(nb)]
    ts = [torch.randint(0, 10, (bs,), generator=gen) for _ in range(nb)]
    # oracle (fp32, CPU): same initial state, same optimizer hyper-parameters
    net = O.OracleNet(ccf=False, head="512", num_classes=10, seed=C.STATE_SEED).train()
    with torch.no_grad():
        for k, v in {**O.init_state(False, C.STATE_SEED, proj_gamma=0.1), **O.init_head_state("512", 10, C.STATE_SEED)}.items():
            name = k.replace(".", "_")
            if hasattr(net, name):
                getattr(net, name).copy_(v)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    crit = torch.nn.CrossEntropyLoss()
    ref = []
    for i in range(steps):
        out = net(xs[i % nb], dropout=False)
        loss = crit(out, ts[i % nb])
        opt.zero_grad(); loss.backward(); opt.step()
        ref.append(float(loss.detach()))
    m = build("512", 10, proj_gamma=0.1).train()
    _no_dropout(m)
    tr = Trainer(m, lr=1e-3)
    hip = [float(tr.step(xs[i % nb].cuda(), ts[i % nb].cuda())) for i in range(steps)]
    ref, hip = np.array(ref), np.array(hip)
    gap = np.abs(hip - ref) / np.abs(ref)
    big = ref > 0.1
    print("trajectory: ref  " + " ".join("%.3f" % v for v in ref))
    print("trajectory: hip  " + " ".join("%.3f" % v for v in hip))
    print("trajectory: max gap (loss > 0.1) %.4f mean gap %.4f, max abs gap %.4f" % (gap[big].max(), gap[big].mean(), np.abs(hip - ref).max()))
    assert np.isfinite(hip).all()
    assert (np.abs(hip - ref) <= np.maximum(6e-2 * np.abs(ref), 0.03)).all(), (hip, ref)
    assert gap[big].mean() <= 3e-2, gap[big].mean()
    if steps >= 20:
        assert hip[-5:].mean() < 0.2 * hip[:5].mean() and ref[-5:].mean() < 0.2 * ref[:5].mean()
    else:
        assert hip[-1] < hip[0] and ref[-1] < ref[0]


def test_native_step_validates_inputs():
    """The autograd-free step hands raw pointers to the launch lists: host / wrong-shape inputs must raise, not fault."""
    from mnasnet_pytorch_amd.train_step import Trainer
    m = build("512", 10).train()
    tr = Trainer(m, lr=1e-3)
    x = torch.randn(2, 3, 64, 64)
    t = torch.tensor([1, 2])
    with pytest.raises(RuntimeError):
        tr.step(x, t.cuda())                      # CPU image
    with pytest.raises(RuntimeError):
        tr.step(x.cuda(), t)                      # CPU target
    with pytest.raises(ValueError):
        tr.step(torch.randn(2, 4, 64, 64).cuda(), t.cuda())     # wrong channel count
    with pytest.raises(ValueError):
        tr.step(torch.randn(3, 64, 64).cuda(), t.cuda())        # wrong rank
    assert np.isfinite(float(tr.step(x.cuda(), t.cuda())))


def test_bench_shape_one_step_finite():
    """bs=32 at 224x224 (the bench shape, smaller batch): one Trainer step runs and the loss is finite."""
    from mnasnet_pytorch_amd.train_step import Trainer
    m = build("512", 1000).train()
    tr = Trainer(m, lr=1e-3)
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(32, 3, 224, 224, device="cuda", generator=g)
    t = torch.randint(0, 1000, (32,), device="cuda", generator=g)
    l0 = float(tr.step(x, t)); l1 = float(tr.step(x, t)); l2 = float(tr.step(x, t))
    assert np.isfinite([l0, l1, l2]).all() and l2 < l0


def test_trainer_steps_bit_reproducible_at_bench_resolution():
    """Two Trainers built from the same state take four optimizer steps on the same 64x3x224x224 batch: losses and final
    parameters are bit-identical (two HIP streams, fused Adam, flat gradient buffer -- no run-to-run variation)."""
    from mnasnet_pytorch_amd.train_step import Trainer
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(64, 3, 224, 224, device="cuda", generator=g)
    target = torch.randint(0, 10, (64,), device="cuda", generator=g)
    runs = []
    for _ in range(2):
        m = build("512", proj_gamma=0.1).train()
        _no_dropout(m)
        t = Trainer(m, lr=1e-3)
        losses = [t.step(x, target).clone() for _ in range(4)]
        torch.cuda.synchronize()
        runs.append((torch.stack(losses), t.flat_p.clone()))
    assert torch.isfinite(runs[0][0]).all()
    assert torch.equal(runs[0][0], runs[1][0]), (runs[0][0], runs[1][0])
    assert torch.equal(runs[0][1], runs[1][1])


# ---- engine / optimizer contract (round-2 advisor findings) ------------------------------------------------------------
def test_dropped_graph_releases_its_program():
    """A train-mode forward whose graph is dropped without backward (exception, LR finder, BN recalibration) hands its
    activation buffers back: repeated forwards reuse ONE program instead of allocating a new set each call."""
    m = build("512").train()
    x = C.det_input((2, 3, 64, 64)).cuda()
    eng = m.features._engine()
    for _ in range(6):
        out = m(x)
        del out
    progs = [p for lst in eng.programs.values() for p in lst]
    assert len(progs) == 1 and not progs[0].busy
    # two graphs alive at once get two programs; both are released by their backward
    a, b = m(x), m(x)
    assert len([p for lst in eng.programs.values() for p in lst]) == 2
    (a.sum() + b.sum()).backward()
    assert not any(p.busy for lst in eng.programs.values() for p in lst)


def test_second_backward_raises():
    m = build("512").train()
    _no_dropout(m)
    out = m(C.det_input((2, 3, 64, 64)).cuda())
    loss = out.sum()
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="ONE backward per forward"):
        loss.backward()


def test_too_many_live_forwards_raise():
    m = build("512").train()
    x = C.det_input((2, 3, 64, 64)).cuda()
    keep = [m(x) for _ in range(4)]
    with pytest.raises(RuntimeError, match="alive at once"):
        m(x)
    del keep


@pytest.mark.parametrize("shape", [(4, 3, 64, 64), (2, 3, 57, 75)])
def test_image_gradient_vs_mirror(shape):
    """dL/d image through the drop-in module (round 5: the stem's input gradient, csrc/mnas_stem.hip k_stem_dgrad; autograd
    completeness -- train.py:427 never asks for it): the front of the network (stem, SepConv, the 112x112 stage's three block
    applications and the stride-2 conv; well-conditioned state) against the bf16 mirror.  The image gradient is an ELEMENTWISE
    function of the stem's dy (32 channels x <= 4 taps per pixel: no averaging over pixels as in a weight gradient), so it carries
    dy's own bf16 noise after 12 layers of backward: measured rel-L2 0.097-0.105, cosine 0.994-0.995 (bounds 0.15 / 0.99) -- the level
    of the PARAMETER gradients of the same pass at these small sizes (median 0.097-0.101, max 0.15-0.16); the kernel itself is held
    to 2e-3 on identical inputs by tests/test_gpu_kernels.py::test_stem.
    The parameter gradients of the same backward must not change with the extra launch."""
    from mnasnet_pytorch_amd import Mnasnet
    from oracle import bf16_mirror as M

    class Front(torch.nn.Module):
        def __init__(self, mods):
            super().__init__()
            self.features = torch.nn.Sequential(*mods)

        def forward(self, x):
            return self.features(x)

    net = Mnasnet(cut_channels_first=False)
    st = O.init_state(False, C.STATE_SEED, proj_gamma=0.1)
    net.load_state_dict(st)
    net = net.cuda().train()
    x0 = C.det_input(shape)
    front = Front([net.features[0], net.features[1], net.features[2]])
    res = []
    for rg in (True, False):
        front.zero_grad(set_to_none=True)
        x = x0.cuda().requires_grad_(rg)
        y = front(x)
        cot = C.cotangent(tuple(y.shape))
        (y * cot.cuda()).sum().backward()
        res.append((x.grad.cpu() if rg else None, {k: p.grad.clone() for k, p in front.named_parameters()}))
    dx, g_with = res[0]
    for k, gv in res[1][1].items():
        assert torch.equal(gv, g_with[k]), k              # same launches, same bits
    prog, _ = O.build_program(False)
    r = M.run(prog[:7], O.init_state(False, C.STATE_SEED, proj_gamma=0.1), x0, True, cot, need_dx=True)
    assert r["dx"] is not None and tuple(r["dx"].shape) == shape
    a, b = dx.double().flatten(), r["dx"].double().flatten()
    e, cos = float((a - b).norm() / b.norm()), float((a @ b) / (a.norm() * b.norm()))
    pe = sorted(float((g_with[k].cpu().double() - r["grads"][k].double()).norm() / (r["grads"][k].double().norm() + 1e-30))
                for k in g_with if not k.endswith("conv.bias"))
    print("image gradient vs mirror: rel-L2 %.4f cosine %.5f; parameter gradients of the same pass: median %.4f max %.4f"
          % (e, cos, pe[len(pe) // 2], pe[-1]))
    assert bool(torch.isfinite(dx).all()) and e < 0.15 and cos > 0.99


def test_graph_replay_is_bit_identical():
    """Engine.use_graphs (launch lists captured into hipGraphs, csrc/mnas_abi.hip mnas_graph_create): the same kernels in the same
    order, so 6 Adam steps must leave bit-identical parameters and losses -- with ONE resident batch (the graphs are captured once
    and replayed) and with batches that alternate between two tensors (a data loader: from the second tensor on every batch is copied
    into a buffer of the program and the graphs are captured once more against that buffer)."""
    from mnasnet_pytorch_amd.train_step import Trainer

    def run(graphs, nb):
        torch.manual_seed(0)
        m = build("512", 10, proj_gamma=0.1).train()
        tr = Trainer(m, lr=1e-3)
        tr.engine.use_graphs = graphs
        gen = torch.Generator().manual_seed(5)
        xs = [torch.randn(8, 3, 64, 64, generator=gen).cuda() for _ in range(nb)]
        ts = [torch.randint(0, 10, (8,), generator=gen).cuda() for _ in range(nb)]
        losses = [float(tr.step(xs[i % nb], ts[i % nb])) for i in range(6)]
        captured = sum(1 for lst in tr.engine.programs.values() for p in lst for s_ in p._graphs.values() if s_["exec"] is not None)
        return losses, tr.flat_p.clone(), captured

    for nb in (1, 2):
        l0, p0, c0 = run(False, nb)
        l1, p1, c1 = run(True, nb)
        assert c0 == 0 and l0 == l1 and torch.equal(p0, p1), (nb, l0, l1)
        assert c1 >= 2                             # the forward list and at least one backward segment live as graphs


def test_image_gradient_whole_model_finite():
    """x.requires_grad_(True) through the whole drop-in model (features + head + loss): x.grad exists, is finite and not zero, and
    the Trainer path (images without grad) is unaffected."""
    m = build("512").train()
    _no_dropout(m)
    x = C.det_input((4, 3, 64, 64)).cuda().requires_grad_(True)
    t = torch.tensor([1, 2, 3, 4]).cuda()
    loss = torch.nn.CrossEntropyLoss()(m(x), t)
    loss.backward()
    assert x.grad is not None and tuple(x.grad.shape) == (4, 3, 64, 64)
    assert bool(torch.isfinite(x.grad).all()) and float(x.grad.abs().max()) > 0


def test_wrong_dtype_and_mixed_modes_fail_loudly():
    x = C.det_input((2, 3, 64, 64)).cuda()
    m = build("512").train()
    m.features.half()
    with pytest.raises(TypeError, match="float32"):
        m.features(x)
    m2 = build("512").train()
    m2.features[0].bn.eval()                     # frozen-BN fine-tuning is not silently ignored
    with pytest.raises(NotImplementedError, match="mixed train/eval"):
        m2(x)


def test_flat_adam_is_a_torch_optimizer():
    """Schedulers drive it, param_groups[0]['lr'] reads back (train.py:450), state_dict round-trips the moments, and a
    frozen parameter is not moved even with weight decay."""
    from mnasnet_pytorch_amd.train_step import FlatAdam, Trainer
    x = C.det_input((4, 3, 64, 64)).cuda()
    target = torch.tensor([1, 3, 5, 7]).cuda()
    m = build("512", proj_gamma=0.1).train(); _no_dropout(m)
    tr = Trainer(m, lr=1e-3, weight_decay=1e-2)
    opt = tr.optimizer
    assert isinstance(opt, torch.optim.Optimizer) and isinstance(opt, FlatAdam)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[1], gamma=0.1)
    tr.step(x, target); sched.step()
    assert abs(opt.state_dict()["param_groups"][0]["lr"] - 1e-4) < 1e-12 and abs(tr.lr - 1e-4) < 1e-12
    plateau = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=0.5, patience=0)
    plateau.step(1.0); plateau.step(2.0)
    assert abs(tr.lr - 5e-5) < 1e-12
    # frozen features: only the head moves
    m.freeze()
    w0 = m.features[0].conv.weight.detach().clone()
    h0 = m.classifier[1].weight.detach().clone()
    tr.step(x, target)
    assert torch.equal(m.features[0].conv.weight.detach(), w0)
    assert not torch.equal(m.classifier[1].weight.detach(), h0)
    m.unfreeze()
    # checkpoint round trip: a second trainer resumed from the state dicts continues bit-identically
    sd_model = {k: v.clone() for k, v in m.state_dict().items()}
    sd_opt = tr.state_dict()
    l_a = float(tr.step(x, target))
    m2 = build("512", proj_gamma=0.1).train(); _no_dropout(m2)
    m2.load_state_dict(sd_model)
    tr2 = Trainer(m2, lr=123.0)
    tr2.load_state_dict(sd_opt)
    assert abs(tr2.lr - tr.lr) < 1e-12 and tr2.step_count == tr.step_count - 1
    l_b = float(tr2.step(x, target))
    assert l_a == l_b
    assert torch.equal(tr.flat_p, tr2.flat_p)


def test_fused_pool_matches_two_step_path():
    """FineTuneModelPool.forward with the global average pool fused into the engine (mnas_pool_act / mnas_pool_bwd) ==
    the two-step path features -> AdaptiveAvgPool2d(1): logits to 1e-5 (fp32 mean, different summation order), every
    gradient to 2e-3 relative L2 (the backward differs only in where g/HW is rounded to bf16)."""
    x = C.det_input((4, 3, 96, 64)).cuda()
    target = torch.tensor([1, 3, 5, 7]).cuda()
    res = []
    for fuse in (True, False):
        m = build("512", proj_gamma=0.1).train(); _no_dropout(m)
        m.fuse_pool = fuse
        out = m(x)
        torch.nn.CrossEntropyLoss()(out, target).backward()
        res.append((out.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}))
    assert rl2(res[0][0].cpu(), res[1][0].cpu()) < 1e-5
    for k in res[0][1]:
        if k.endswith("conv.bias"):
            continue
        a, b = res[0][1][k], res[1][1][k]
        assert rl2(a.cpu(), b.cpu()) < 2e-3 or float((a - b).abs().max()) < 1e-7, k
    # eval mode too
    m = build("512").eval()
    with torch.no_grad():
        m.fuse_pool = True
        a = m(x)
        m.fuse_pool = False
        b = m(x)
    assert rl2(a.cpu(), b.cpu()) < 1e-5


def test_input_pipeline_on_device_normalisation_and_uint8():
    """SURVEY 8(f) row 4, input half: the dataset's transforms.Normalize(mean, std) (datasets.py:474-516, constants
    classifiers.py:91-92) fused into the stem conv's load, for float [0,1] and raw uint8 images.  Reference semantics: the model
    sees (u8 / 255 - mean) / std.  Three routes must agree: normalised floats computed on the host (the reference's route), float
    [0,1] images normalised on load, uint8 images normalised on load -- logits <= 2e-2 relative L2 (bf16 rounding of a value
    computed as fma on load vs two fp32 ops on the host), stem weight gradients <= 2e-2."""
    g = torch.Generator().manual_seed(5)
    u8 = torch.randint(0, 256, (8, 3, 64, 64), generator=g, dtype=torch.uint8)
    t = torch.tensor([1, 3, 5, 7, 0, 2, 4, 6]).cuda()
    crit = torch.nn.CrossEntropyLoss()
    outs = []
    for route in ("host", "float", "uint8"):
        m = build("512", 10, proj_gamma=0.1).train()
        _no_dropout(m)
        mean = torch.tensor(m.mean).view(1, 3, 1, 1)
        std = torch.tensor(m.std).view(1, 3, 1, 1)
        if route == "host":
            x = ((u8.float() / 255.0 - mean) / std).cuda()
        elif route == "float":
            m.normalize_on_device()
            x = (u8.float() / 255.0).cuda()
        else:
            m.normalize_on_device()
            x = u8.cuda()
        out = m(x)
        loss = crit(out, t)
        loss.backward()
        outs.append((out.detach().cpu(), m.features[0].conv.weight.grad.cpu().clone(), float(loss)))
    for o, gw, l in outs[1:]:
        assert rl2(o, outs[0][0]) < 2e-2 and rl2(gw, outs[0][1]) < 2e-2, (rl2(o, outs[0][0]), rl2(gw, outs[0][1]))
        assert abs(l - outs[0][2]) < 1e-2 * abs(outs[0][2])
    # the autograd-free Trainer step takes uint8 batches too
    from mnasnet_pytorch_amd.train_step import Trainer
    m = build("512", 10, proj_gamma=0.1).train()
    _no_dropout(m)
    m.normalize_on_device()
    tr = Trainer(m, lr=1e-3)
    l0 = float(tr.step(u8.cuda(), t)); l1 = float(tr.step(u8.cuda(), t))
    assert abs(l0 - outs[0][2]) < 1e-2 * abs(l0) and l1 < l0
    # without the transform a uint8 batch is converted with .float() (train.py:427), nothing else
    m2 = build("512", 10, proj_gamma=0.1).eval()
    with torch.no_grad():
        assert rl2(m2(u8.cuda()).cpu(), m2(u8.float().cuda()).cpu()) < 1e-6


@pytest.mark.parametrize("kind,kw", [("rmsprop", {}), ("rmsprop", {"momentum": 0.9}), ("sgd", {}), ("sgd", {"momentum": 0.9}),
                                     ("sgd", {"momentum": 0.9, "nesterov": True}), ("adam", {})])
def test_flat_optimizers_match_torch(kind, kw):
    """train.py:218-231 offers Adam, RMSprop and SGD (each constructed with lr only).  The fused flat-buffer updates
    (mnas_adam_step / mnas_rmsprop_step / mnas_sgd_step) against the stock torch optimizers on the same gradient sequence, five
    steps with weight decay: parameters equal to 1e-6 relative (fp32 arithmetic, same formulas)."""
    from mnasnet_pytorch_amd import _lib as Lb
    from mnasnet_pytorch_amd.train_step import FlatAdam, FlatRMSprop, FlatSGD
    n = 10007
    g0 = torch.Generator().manual_seed(7)
    p0 = torch.randn(n, generator=g0)
    grads = [torch.randn(n, generator=g0) * (0.5 + 0.2 * i) for i in range(5)]
    ref = torch.nn.Parameter(p0.clone())
    cls = {"adam": torch.optim.Adam, "rmsprop": torch.optim.RMSprop, "sgd": torch.optim.SGD}[kind]
    opt_ref = cls([ref], lr=1e-2, weight_decay=1e-2, **kw)
    flat_p, flat_g = p0.clone().cuda(), torch.zeros(n, device="cuda")
    par = torch.nn.Parameter(flat_p.view(n))
    par.data = flat_p.view(n)
    fcls = {"adam": FlatAdam, "rmsprop": FlatRMSprop, "sgd": FlatSGD}[kind]
    opt = fcls([par], flat_p, flat_g, lr=1e-2, weight_decay=1e-2, **kw)
    assert isinstance(opt, torch.optim.Optimizer)
    for g in grads:
        ref.grad = g.clone()
        opt_ref.step()
        flat_g.copy_(g.cuda())
        opt.step()
    torch.cuda.synchronize()
    assert rl2(flat_p.cpu(), ref.detach()) < 1e-6, (kind, kw)
    # the Trainer builds them by name, like train.py's --optimizer flag
    from mnasnet_pytorch_amd.train_step import Trainer
    m = build("512", 10, proj_gamma=0.1).train()
    _no_dropout(m)
    tr = Trainer(m, lr=1e-3, optimizer=kind, **kw)
    x = C.det_input((4, 3, 64, 64)).cuda()
    t = torch.tensor([1, 3, 5, 7]).cuda()
    losses = [float(tr.step(x, t)) for _ in range(3)]
    assert np.isfinite(losses).all() and type(tr.optimizer) is fcls
    with pytest.raises(ValueError):
        Trainer(build("512", 10).train(), optimizer="adagrad")


def test_on_device_normalisation_survives_deepcopy_and_eval_fused_block():
    """normalize_on_device() is MODULE state (the engine is rebuilt lazily after a copy and must pick it up again)."""
    import copy
    g = torch.Generator().manual_seed(9)
    u8 = torch.randint(0, 256, (4, 3, 224, 224), generator=g, dtype=torch.uint8).cuda()
    m = build("512", 10, proj_gamma=0.1).eval()
    m.normalize_on_device()
    with torch.no_grad():
        y0 = m(u8)
        m2 = copy.deepcopy(m)
        assert m2.input_norm_on_device and rl2(m2(u8).cpu(), y0.cpu()) < 1e-6
