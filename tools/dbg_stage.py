"""Debug helper: repeatability of one stand-alone stage against the bf16 mirror (prints relative L2 errors)."""
import os, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import cases as C
from cases import O
from oracle import bf16_mirror as M
import mnasnet_pytorch_amd._lib as _L
if os.environ.get('DBG_LIB'):
    _L.LIB_PATH = os.environ['DBG_LIB']
from mnasnet_pytorch_amd import MBConv
def rl2(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))
name = sys.argv[1] if len(sys.argv) > 1 else 'stage_16_24_ccfF'
cin, cout, t, layers, k, reduce, ccf, N, H, W = C.STAGES[name]
def fill(module, prefix, seed=C.STATE_SEED):
    sd = module.state_dict()
    new, first = {}, {}
    for k, v in sd.items():
        src = first.setdefault(v.data_ptr(), k) if v.dim() > 0 else k
        new[k] = O.det_param(prefix + "." + src, tuple(v.shape), seed).to(v.dtype)
    module.load_state_dict(new)
outs = []
for rep in range(4):
    m = MBConv(cin, cout, t, layers, kernel_size=k, reduce=reduce, cut_channels_first=ccf)
    fill(m, name)
    m = m.cuda().train()
    x0 = C.det_input((N, cin, H, W))
    x = x0.cuda().requires_grad_(True)
    y = m(x)
    cot = C.cotangent(tuple(y.shape))
    (y * cot.cuda()).sum().backward()
    outs.append((y.detach().cpu(), x.grad.cpu()))
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
prog = ([("conv", conv)] if ccf else []) + [("block", blk)] * layers + ([] if ccf else [("conv", conv)])
r = M.run(prog, st, x0, True, cot, need_dx=True)
for y, dx in outs:
    print("y vs mirror %.5f   dx vs mirror %.5f   dx vs run0 %.5f" % (rl2(y, r["y"]), rl2(dx, r["dx"]), rl2(dx, outs[0][1])))
