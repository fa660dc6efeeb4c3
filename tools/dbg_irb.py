import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
import torch
import cases as C
from test_gpu_model import _stage_setup, FULL_STAGES
name = "features7_192_320_k3_7"
junk = [torch.full((64 * 1024 * 1024,), float("nan"), device="cuda") for _ in range(8)]      # poison the allocator's free blocks
del junk
for N, mode in ((32, "full"), (32, False), (32, "fwd"), (30, False), (64, False)):
    spec = FULL_STAGES[name][:7] + (N,) + FULL_STAGES[name][8:]
    m, _, _, shp = _stage_setup(name, 0.1, spec)
    m._engine().fuse_irb = mode
    x = C.det_input(shp).cuda().requires_grad_(True)
    y = m(x)
    cot = C.cotangent(tuple(y.shape)).cuda()
    (y * cot).sum().backward()
    torch.cuda.synchronize()
    bad = [kk for kk, p in m.named_parameters() if not bool(torch.isfinite(p.grad).all())]
    print(N, mode, "non-finite grads:", bad, "dx finite", bool(torch.isfinite(x.grad).all()))
    if bad:
        g = dict(m.named_parameters())[bad[0]].grad
        nz = (~torch.isfinite(g)).nonzero()
        print("   count", nz.shape[0], "of", g.numel(), "first", nz[:5].tolist(), "rows", sorted(set(nz[:, 0].tolist()))[:20])

# where does the NaN sit in the P partial table?
N = 32
spec = FULL_STAGES[name][:7] + (N,) + FULL_STAGES[name][8:]
junk = [torch.full((64 * 1024 * 1024,), float("nan"), device="cuda") for _ in range(8)]
del junk
m, _, _, shp = _stage_setup(name, 0.1, spec)
eng = m._engine()
eng.fuse_irb = "full"
x = C.det_input(shp).cuda().requires_grad_(True)
y = m(x)
(y * C.cotangent(tuple(y.shape)).cuda()).sum().backward()
torch.cuda.synchronize()
E_, C_ = 1152, 192
for nm in ("scratch_wgrad2", "scratch_wgrad3", "scratch_wgrad4"):
    t = getattr(eng, nm)
    for nparts in (8,):
        v = t[:nparts * E_ * C_].view(nparts, E_, C_)
        bad = (~torch.isfinite(v)).nonzero()
        print(nm, "P-view non-finite:", bad.shape[0], bad[:6].tolist(), "parts", sorted(set(bad[:, 0].tolist())), "cols", sorted(set(bad[:, 2].tolist()))[:10])
