"""PCIe-inclusive step rate: the reference hands a HOST fp32 batch to the step (train.py:427
`input.float().to(device)`).  Measures (a) serial copy + step, (b) the next batch's copy on a second stream under the step."""
import sys, time, torch, contextlib, io
sys.path.insert(0, '.')
from mnasnet_pytorch_amd import FineTuneModelPool, load_model
from mnasnet_pytorch_amd.train_step import Trainer
dev = torch.device("cuda", 0)
with contextlib.redirect_stdout(io.StringIO()):
    base = load_model("mnasnet")
model = FineTuneModelPool(base, "mnasnet", 1000, "512").to(dev).train()
tr = Trainer(model, lr=1e-3)
B = 256
xh = [torch.randn(B, 3, 224, 224).pin_memory() for _ in range(2)]
th = torch.randint(0, 1000, (B,)).pin_memory()
xd = [torch.empty(B, 3, 224, 224, device=dev) for _ in range(2)]
td = th.to(dev)
for _ in range(5): tr.step(xd[0], td)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for i in range(n):
    xd[0].copy_(xh[i & 1], non_blocking=True)
    tr.step(xd[0], td)
torch.cuda.synchronize(); ser = (time.perf_counter() - t0) / n
cs = torch.cuda.Stream()
ev = [torch.cuda.Event(), torch.cuda.Event()]
done = [torch.cuda.Event(), torch.cuda.Event()]
with torch.cuda.stream(cs):
    xd[0].copy_(xh[0], non_blocking=True); ev[0].record(cs)
t0 = time.perf_counter()
for i in range(n):
    cur, nxt = i & 1, (i + 1) & 1
    with torch.cuda.stream(cs):
        if i >= 1: cs.wait_event(done[nxt])              # the step that read xd[nxt] has finished
        xd[nxt].copy_(xh[nxt], non_blocking=True); ev[nxt].record(cs)
    torch.cuda.current_stream().wait_event(ev[cur])
    tr.step(xd[cur], td)
    done[cur].record()
torch.cuda.synchronize(); ovl = (time.perf_counter() - t0) / n
for _ in range(3): tr.step(xd[0], td)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): tr.step(xd[0], td)
torch.cuda.synchronize(); res = (time.perf_counter() - t0) / n
print("resident %.2f ms (%.0f img/s) | serial H2D+step %.2f ms (%.0f img/s) | copy overlapped %.2f ms (%.0f img/s)" % (
    res * 1e3, B / res, ser * 1e3, B / ser, ovl * 1e3, B / ovl))
