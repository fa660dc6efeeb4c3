"""Does averaging over K independent inputs make a per-tensor SCALE check of the whole-network gradients tight?  K runs (different
input / cotangent seeds) of N images at 224x224; per tensor the projection coefficient of the SUMMED gradients
<sum_k g_hip, sum_k g_mirror> / |sum_k g_mirror|^2 ... and of the per-run mean.     python3 tools/probe/scale_noise_k.py [K=6] [N=4] [pg=0.1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
torch.set_num_threads(16)
import cases as C
from oracle import mnasnet_oracle as O, bf16_mirror as M
from mnasnet_pytorch_amd import Mnasnet
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
pg = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
for ccf in (False, True):
    prog, _ = O.build_program(ccf)
    projs = {}
    for k in range(K):
        m = Mnasnet(cut_channels_first=ccf)
        m.load_state_dict(O.init_state(ccf, C.STATE_SEED, proj_gamma=pg))
        m = m.cuda().train()
        x0 = C.det_input((N, 3, 224, 224), seed=C.INPUT_SEED + 101 * k)
        y = m(x0.cuda())
        cot = C.cotangent(tuple(y.shape), seed=C.COT_SEED + 77 * k)
        (y * cot.cuda()).sum().backward()
        grads = {kk: p.grad.cpu() for kk, p in m.named_parameters()}
        r = M.run(prog, O.init_state(ccf, C.STATE_SEED, proj_gamma=pg), x0, True, cot)
        for kk, gv in grads.items():
            if kk.endswith("conv.bias"):
                continue
            a, b = gv.double().flatten(), r["grads"][kk].double().flatten()
            projs.setdefault(kk, []).append(float(a @ b) / float(b @ b))
        del m, y
        torch.cuda.empty_cache()
    print("ccf=%s pg=%g K=%d N=%d" % (ccf, pg, K, N))
    for sfx in ("bn.weight", "bn.bias", "conv.weight"):
        sel = {kk: np.array(v) for kk, v in projs.items() if kk.endswith(sfx)}
        single = np.concatenate([np.abs(v - 1) for v in sel.values()])
        means = np.array([abs(v.mean() - 1) for v in sel.values()])
        meds = np.array([abs(np.median(v) - 1) for v in sel.values()])
        print("   %-11s single-run |proj-1| worst %.4f | mean over K worst %.4f | median over K worst %.4f" % (sfx, single.max(), means.max(), meds.max()), flush=True)
