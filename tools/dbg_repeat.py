"""Debug helper: run the forward of one stand-alone stage repeatedly and report which program tensor first differs."""
import os, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import cases as C
from cases import O
from mnasnet_pytorch_amd import MBConv
name = sys.argv[1] if len(sys.argv) > 1 else 'stage_16_24_ccfF'
cin, cout, t, layers, k, reduce, ccf, N, H, W = C.STAGES[name]
def fill(module, prefix, seed=C.STATE_SEED):
    sd = module.state_dict()
    new, first = {}, {}
    for kk, v in sd.items():
        src = first.setdefault(v.data_ptr(), kk) if v.dim() > 0 else kk
        new[kk] = O.det_param(prefix + "." + src, tuple(v.shape), seed).to(v.dtype)
    module.load_state_dict(new)
m = MBConv(cin, cout, t, layers, kernel_size=k, reduce=reduce, cut_channels_first=ccf)
fill(m, name)
m = m.cuda().train()
x = C.det_input((N, cin, H, W)).cuda()
snaps = []
for rep in range(8):
    torch.cuda.synchronize()
    xx = x.clone().requires_grad_(True)
    m.zero_grad(set_to_none=True)
    y = m(xx)
    (y * C.cotangent(tuple(y.shape)).cuda()).sum().backward()
    torch.cuda.synchronize()
    eng = m._engine()
    progs = [p for lst in eng.programs.values() for p in lst]
    prog = progs[0]
    snaps.append([t_.clone() for t_ in prog.keep] + [xx.grad.clone()] + [p_.grad.clone() for p_ in m.parameters()])
print("programs", len(progs), "keep", len(snaps[0]))
for rep in range(1, len(snaps)):
    msg = []
    for i, (a, b) in enumerate(zip(snaps[0], snaps[rep])):
        if a.dtype == torch.uint8:
            ne = int((a != b).sum())
        else:
            ne = int((a.float() != b.float()).sum())
        if ne:
            d = float((a.float() - b.float()).abs().max())
            msg.append("#%d%s%s ne=%d max=%.3g" % (i, tuple(a.shape), str(a.dtype)[6:], ne, d))
    print("rep", rep, "; ".join(msg[:6]) if msg else "identical")
