"""How tight can a per-tensor scale check of the whole-network gradients be?    For several conditioning states
(proj_gamma = the BatchNorm weight scale of every projection conv) at 224x224: per-parameter rel-L2, projection coefficient
<g_hip, g_mirror> / |g_mirror|^2 and norm ratio of the HIP engine's gradients against the bf16 mirror's.
    python3 tools/probe/scale_noise.py [N=16] [pg list, e.g. 0.1,0.03,0.01]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
torch.set_num_threads(16)
import cases as C
from oracle import mnasnet_oracle as O, bf16_mirror as M
from mnasnet_pytorch_amd import Mnasnet
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
pgs = [float(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0.1,0.03,0.01").split(",")]
for ccf in (False, True):
    for pg in pgs:
        m = Mnasnet(cut_channels_first=ccf)
        m.load_state_dict(O.init_state(ccf, C.STATE_SEED, proj_gamma=pg))
        m = m.cuda().train()
        x0 = C.det_input((N, 3, 224, 224))
        y = m(x0.cuda())
        cot = C.cotangent(tuple(y.shape))
        (y * cot.cuda()).sum().backward()
        grads = {kk: p.grad.cpu() for kk, p in m.named_parameters()}
        y_ = y.detach().cpu()
        prog, _ = O.build_program(ccf)
        r = M.run(prog, O.init_state(ccf, C.STATE_SEED, proj_gamma=pg), x0, True, cot)
        ey = float((y_ - r["y"]).norm() / r["y"].norm())
        rows = {}
        for kk, gv in grads.items():
            if kk.endswith("conv.bias"):
                continue
            a, b = gv.double().flatten(), r["grads"][kk].double().flatten()
            rows[kk] = (float((a - b).norm() / b.norm()), float(a @ b) / float(b @ b), float(a.norm() / b.norm()), a.numel())
        print("ccf=%s proj_gamma=%g N=%d: y rel-L2 %.4f" % (ccf, pg, N, ey))
        for sfx in ("bn.weight", "bn.bias", "conv.weight"):
            sel = [v for kk, v in rows.items() if kk.endswith(sfx)]
            print("   %-11s rel-L2 median %.4f max %.4f | projection [%.4f, %.4f] | norm ratio [%.4f, %.4f]"
                  % (sfx, float(np.median([v[0] for v in sel])), max(v[0] for v in sel), min(v[1] for v in sel), max(v[1] for v in sel),
                     min(v[2] for v in sel), max(v[2] for v in sel)), flush=True)
        del m, y
        torch.cuda.empty_cache()
