"""Debug helper: run one full-size stage forward+backward and report which program tensors / gradients contain NaN."""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import cases as C
from cases import O
from mnasnet_pytorch_amd import MBConv
from mnasnet_pytorch_amd import _lib as L
spec = (96, 192, 6, 4, 5, True, False, int(sys.argv[1]) if len(sys.argv) > 1 else 256, 14, 14)
cin, cout, t, layers, k, reduce, ccf, N, H, W = spec
m = MBConv(cin, cout, t, layers, kernel_size=k, reduce=reduce, cut_channels_first=ccf).cuda().train()
if len(sys.argv) > 2:
    m._engine().pw_fused_min_pixels = int(sys.argv[2])
x = C.det_input((N, cin, H, W)).cuda().requires_grad_(True)
y = m(x)
eng = m._engine()
prog = [p for lst in eng.programs.values() for p in lst][0]
(y * C.cotangent(tuple(y.shape)).cuda()).sum().backward()
torch.cuda.synchronize()
print("y nan", bool(torch.isnan(y).any()), "dx nan", bool(torch.isnan(x.grad).any()), "dx nan count", int(torch.isnan(x.grad).sum()))
for kk, p in m.named_parameters():
    if torch.isnan(p.grad).any():
        print("param grad NaN:", kk, int(torch.isnan(p.grad).sum()), "of", p.grad.numel())
for i, t_ in enumerate(prog.keep):
    if t_.dtype in (torch.bfloat16, torch.float32) and torch.isnan(t_.float()).any():
        print("keep #%d %s %s nan=%d" % (i, tuple(t_.shape), str(t_.dtype)[6:], int(torch.isnan(t_.float()).sum())))
for i, t_ in enumerate(prog.keep[:40]):
    print("keep", i, tuple(t_.shape), str(t_.dtype)[6:])
# BN-backward partial tables still in scratch: any NaN?
print("scratch_red nan", int(torch.isnan(eng.scratch_red).sum()), "scratch_stats nan", int(torch.isnan(eng.scratch_stats).sum()))
