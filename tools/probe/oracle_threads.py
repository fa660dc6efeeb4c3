import sys, time, torch
sys.path.insert(0, 'tests/golden'); sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import cases as C
from cases import O
print("default threads", torch.get_num_threads())
for thr in (None, 64, 32, 16, 8):
    if thr: torch.set_num_threads(thr)
    for (bs, size) in ((24, 64), (32, 224)):
        gen = torch.Generator().manual_seed(123)
        x = torch.randn(bs, 3, size, size, generator=gen); t = torch.randint(0, 10, (bs,), generator=gen)
        net = O.OracleNet(ccf=False, head="512", num_classes=10, seed=C.STATE_SEED).train()
        opt = torch.optim.Adam(net.parameters(), lr=1e-3); crit = torch.nn.CrossEntropyLoss()
        for i in range(4):
            if i == 1: t0 = time.time()
            loss = crit(net(x, dropout=False), t); opt.zero_grad(); loss.backward(); opt.step()
        print("threads", torch.get_num_threads(), (bs, size), "%.3f s/step" % ((time.time() - t0) / 3), flush=True)
