"""Soak / race check at the bench shape: the SAME K-step training run (bs 256, 224x224, head '512', 1000 classes, Adam, dropout
on, 4 fixed batches cycling) executed twice from the same state; every step's loss and the final flat parameter / moment buffers
must be bit-identical (no float atomics, fixed summation orders: a difference is a race or an uninitialised read -- the kind of
bug that never shows in small tests, cf. dma_barrier in csrc/mnas_common.h), the loss finite throughout and lower at the end.
usage (GPU box): python3 tools/soak.py [steps=150] [batch=256] [se|ccf|graphs|HxW ...]   (se: the 5x5 + squeeze-excite variant of BASELINE
config 4; ccf: Mnasnet(cut_channels_first=True); graphs: the second run replays hipGraphs)"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mnasnet_pytorch_amd import FineTuneModelPool, Mnasnet, load_model
from mnasnet_pytorch_amd.train_step import Trainer

K = int(sys.argv[1]) if len(sys.argv) > 1 else 150
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
SE = "se" in sys.argv[3:]
CCF = "ccf" in sys.argv[3:]             # the reference's default topology, Mnasnet(cut_channels_first=True)
GRAPHS = "graphs" in sys.argv[3:]       # second run replays the launch lists as hipGraphs (Engine.use_graphs): must give the same bits
HW = (224, 224)
for a_ in sys.argv[3:]:
    if "x" in a_ and a_.replace("x", "").isdigit():
        HW = tuple(int(v) for v in a_.split("x"))            # e.g. 384x512: a rectangular cluster of BASELINE config 5


def run(graphs=False):
    torch.manual_seed(1234)
    base = Mnasnet(False, kernel_size=5, se_ratio=0.25) if SE else (Mnasnet(cut_channels_first=True) if CCF else load_model("mnasnet"))
    m = FineTuneModelPool(base, "mnasnet", 1000, "512").cuda().train()
    tr = Trainer(m, lr=1e-3)
    tr.engine.use_graphs = graphs
    g = torch.Generator(device="cuda").manual_seed(7)
    xs = [torch.randn(B, 3, HW[0], HW[1], device="cuda", generator=g) for _ in range(4)]
    ts = [torch.randint(0, 1000, (B,), device="cuda", generator=g) for _ in range(4)]
    losses = []
    for i in range(K):
        losses.append(float(tr.step(xs[i % 4], ts[i % 4])))
    torch.cuda.synchronize()
    digest = hashlib.sha256(tr.flat_p.cpu().numpy().tobytes()).hexdigest()
    bn = hashlib.sha256(torch.cat([b.flatten().float() for b in m.buffers()]).cpu().numpy().tobytes()).hexdigest()
    fin = bool(torch.isfinite(tr.flat_p).all())
    del tr, m
    torch.cuda.empty_cache()
    return losses, digest, bn, fin


a = run()
b = run(GRAPHS)
print("steps %d, batch %d: loss %.4f -> %.4f (min %.4f)" % (K, B, a[0][0], a[0][-1], min(a[0])))
print("run 1 params %s  buffers %s" % (a[1][:16], a[2][:16]))
print("run 2 params %s  buffers %s" % (b[1][:16], b[2][:16]))
bad = [i for i, (u, v) in enumerate(zip(a[0], b[0])) if u != v]
ok = (not bad) and a[1] == b[1] and a[2] == b[2] and a[3] and all(x == x and abs(x) < 1e4 for x in a[0]) and \
    sum(a[0][-8:]) < sum(a[0][:8])
print("first differing step:", bad[0] if bad else None)
print("SOAK", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
