"""Generates tests/golden/*.npz by running the REFERENCE implementation (imported from
/root/reference/src, which exists only in the build container).  Run:  python tests/golden/make_golden.py

The fixtures hold inputs' seeds and the reference's OUTPUTS only.  Weights/inputs are the closed-form
fills of oracle.mnasnet_oracle (det_param / det_uniform), loaded into the reference modules via
load_state_dict, so nothing of the reference travels.  torch.set_num_threads(1) for reproducibility
(CPU results differ bitwise with thread count)."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases as C  # noqa: E402
from cases import O  # noqa: E402

REF = "/root/reference/src"
sys.path.insert(0, REF)
# classifiers.py imports torchvision at module scope (classifiers.py:5) but the mnasnet branch never
# touches it; torchvision is not installed here -> stub module object.
if "torchvision" not in sys.modules:
    tv = types.ModuleType("torchvision")
    tv.models = types.ModuleType("torchvision.models")
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tv.models
from models import mnasnet as R  # noqa: E402
from models import classifiers as RC  # noqa: E402

torch.set_num_threads(1)


def npy(t):
    return t.detach().cpu().numpy()


def load_det(module, prefix, seed=C.STATE_SEED):
    """Fill a reference module with det_param values keyed '<prefix>.<name>'."""
    sd = module.state_dict()
    new, first = {}, {}
    for k, v in sd.items():
        # aliases of a shared block (list-multiply, mnasnet.py:162-164) get the FIRST alias's value
        src = first.setdefault(v.data_ptr(), k) if v.dim() > 0 else k
        new[k] = O.det_param(prefix + "." + src, tuple(v.shape), seed).to(v.dtype)
    module.load_state_dict(new)


def grads_of(module):
    return {k: npy(p.grad) for k, p in module.named_parameters()}


def gen_primitives():
    out = {}
    for name, (cin, cout, k, s, p, g, N, H, W) in C.PRIMITIVES.items():
        for train in (True, False):
            m = R.ConvBlock(cin, cout, kernel_size=k, stride=s, padding=p, groups=g)
            load_det(m, name)
            m.train(train)
            x = C.det_input((N, cin, H, W)).requires_grad_(True)
            y = m(x)
            (y * C.cotangent(tuple(y.shape))).sum().backward()
            tag = name + ("/train" if train else "/eval")
            out[tag + "/y"] = npy(y)
            out[tag + "/dx"] = npy(x.grad)
            for kk, gv in grads_of(m).items():
                out[tag + "/d_" + kk] = gv
            for kk, v in m.state_dict().items():
                if "running" in kk or "tracked" in kk:
                    out[tag + "/" + kk] = npy(v)
    np.savez_compressed(os.path.join(HERE, "primitives.npz"), **out)


def gen_blocks():
    out = {}
    for name, (c, t, k, N, H, W) in C.BLOCKS.items():
        m = R.MBConv_block(c, t, k)
        load_det(m, name)
        m.train()
        x = C.det_input((N, c, H, W)).requires_grad_(True)
        y = m(x)
        (y * C.cotangent(tuple(y.shape))).sum().backward()
        out[name + "/y"] = npy(y)
        out[name + "/dx"] = npy(x.grad)
        for kk, gv in grads_of(m).items():
            out[name + "/d_" + kk] = gv
        for kk, v in m.state_dict().items():
            if "running" in kk or "tracked" in kk:
                out[name + "/" + kk] = npy(v)
    np.savez_compressed(os.path.join(HERE, "blocks.npz"), **out)


def gen_stages():
    out = {}
    for name, (cin, cout, t, layers, k, reduce, ccf, N, H, W) in C.STAGES.items():
        m = R.MBConv(cin, cout, t, layers, kernel_size=k, reduce=reduce, cut_channels_first=ccf)
        load_det(m, name)
        m.train()
        x = C.det_input((N, cin, H, W)).requires_grad_(True)
        y = m(x)
        (y * C.cotangent(tuple(y.shape))).sum().backward()
        out[name + "/y"] = npy(y)
        out[name + "/dx"] = npy(x.grad)
        for kk, gv in grads_of(m).items():      # named_parameters dedups shared -> first alias only
            out[name + "/d_" + kk] = gv
        for kk, v in m.state_dict().items():
            if "running" in kk or "tracked" in kk:
                out[name + "/" + kk] = npy(v)
    np.savez_compressed(os.path.join(HERE, "stages.npz"), **out)


def gen_sepconvs():
    """SepConv (mnasnet.py:64-103) on its own: the network's instance, and repeat = 1 / 2 (the list-multiplied pair)."""
    out = {}
    for name, (cin, cout, k, reduce, repeat, N, H, W) in C.SEPCONVS.items():
        for train in (True, False):
            m = R.SepConv(cin, cout, kernel_size=k, reduce=reduce, repeat=repeat)
            load_det(m, name)
            m.train(train)
            x = C.det_input((N, cin, H, W)).requires_grad_(True)
            y = m(x)
            (y * C.cotangent(tuple(y.shape))).sum().backward()
            tag = name + ("/train" if train else "/eval")
            out[tag + "/y"] = npy(y)
            out[tag + "/dx"] = npy(x.grad)
            for kk, gv in grads_of(m).items():      # named_parameters dedups the shared pair -> first alias only
                out[tag + "/d_" + kk] = gv
            for kk, v in m.state_dict().items():
                if "running" in kk or "tracked" in kk:
                    out[tag + "/" + kk] = npy(v)
    np.savez_compressed(os.path.join(HERE, "sepconvs.npz"), **out)


def gen_nets():
    out = {}
    for name, (ccf, N, H, W, train, pg) in C.NETS.items():
        m = R.Mnasnet(cut_channels_first=ccf)
        m.load_state_dict(O.init_state(ccf, C.STATE_SEED, proj_gamma=pg))
        m.train(train)
        x = C.det_input((N, 3, H, W))
        if train:
            y = m(x)
            (y * C.cotangent(tuple(y.shape))).sum().backward()
            out[name + "/y"] = npy(y)
            for kk, p in m.named_parameters():
                out[name + "/gsum/" + kk] = np.array(C.summarize(p.grad))
            # full gradients for a handful of small tensors (first/last stage, one shared block)
            for kk, p in m.named_parameters():
                if kk.startswith("features.0.") or kk.startswith("features.1.") or ".bn." in kk and kk.startswith("features.2."):
                    out[name + "/g/" + kk] = npy(p.grad)
            for kk, v in m.state_dict().items():
                if "running" in kk:
                    out[name + "/ssum/" + kk] = np.array(C.summarize(v))
                if "tracked" in kk:
                    out[name + "/" + kk] = npy(v)
        else:
            with torch.no_grad():
                y = m(x)
            out[name + "/y"] = npy(y)
            out[name + "/ysum"] = np.array(C.summarize(y))
    np.savez_compressed(os.path.join(HERE, "nets.npz"), **out)


def gen_keys():
    with open(os.path.join(HERE, "state_dict_keys.txt"), "w") as f:
        for ccf in (True, False):
            m = R.Mnasnet(cut_channels_first=ccf)
            f.write("# cut_channels_first=%s\n" % ccf)
            for k, v in m.state_dict().items():
                f.write("%s %s %s\n" % (k, "x".join(map(str, v.shape)) or "scalar", str(v.dtype).replace("torch.", "")))


def gen_heads():
    """FineTuneModelPool over load_model('mnasnet') (classifiers.py:7-17,19-111): eval forward for the four
    head configs, and two Adam train steps (train.py:423-440) with the Dropout probabilities forced to 0 at
    run time (dropout draws torch RNG; the step arithmetic is what is pinned)."""
    out = {}
    num_classes = 10
    for cfg in C.HEADS:
        base = RC.load_model("mnasnet")
        model = RC.FineTuneModelPool(base, "mnasnet", num_classes, cfg)
        fsd = {"features." + k[len("features."):]: v for k, v in O.init_state(False, C.STATE_SEED).items()}
        hsd = O.init_head_state(cfg, num_classes, C.STATE_SEED)
        model.load_state_dict({**fsd, **hsd})
        model.eval()
        x = C.det_input((2, 3, 64, 64))
        with torch.no_grad():
            out["head_%s/eval_logits" % cfg] = npy(model(x))
    # train steps, head '512', dropout p=0
    base = RC.load_model("mnasnet")
    model = RC.FineTuneModelPool(base, "mnasnet", num_classes, "512")
    fsd = {k: v for k, v in O.init_state(False, C.STATE_SEED).items()}
    model.load_state_dict({**fsd, **O.init_head_state("512", num_classes, C.STATE_SEED)})
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    model.train()
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=1e-3)   # train.py:219-221
    crit = torch.nn.CrossEntropyLoss()                                                         # train.py:277
    x = C.det_input((4, 3, 64, 64))
    target = torch.tensor([1, 3, 5, 7])
    losses = []
    for _ in range(2):
        out_ = model(x.float())
        loss = crit(out_, target)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    out["step/losses"] = np.array(losses)
    for kk, p in model.named_parameters():
        out["step/psum/" + kk] = np.array(C.summarize(p))
    np.savez_compressed(os.path.join(HERE, "heads.npz"), **out)


if __name__ == "__main__":
    gen_keys()
    gen_primitives()
    gen_blocks()
    gen_stages()
    gen_sepconvs()
    gen_nets()
    gen_heads()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz") or f.endswith(".txt"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
