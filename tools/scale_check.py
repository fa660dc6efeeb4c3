"""Scale sanity: a few train steps of configurations other than the bench default (ccf=True, rectangular crops, other batch
sizes, eval mode); prints ms/step and the loss, fails on any launch error or non-finite loss."""
import sys, time, torch, contextlib, io
sys.path.insert(0, '.')
from mnasnet_pytorch_amd import FineTuneModelPool, Mnasnet
from mnasnet_pytorch_amd.train_step import Trainer
dev = torch.device("cuda", 0)
for (ccf, B, H, W) in [(False, 64, 384, 512), (True, 256, 224, 224), (False, 100, 224, 224), (True, 48, 512, 384), (False, 7, 97, 131)]:
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        base = Mnasnet(cut_channels_first=ccf)
    model = FineTuneModelPool(base, "mnasnet", 1000, "512").to(dev).train()
    tr = Trainer(model, lr=1e-3)
    x = torch.randn(B, 3, H, W, device=dev); y = torch.randint(0, 1000, (B,), device=dev)
    for _ in range(3): loss = tr.step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): loss = tr.step(x, y)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    lv = float(loss)
    assert lv == lv and abs(lv) < 1e4, lv
    model.eval()
    with torch.no_grad():
        out = model(x)
    assert torch.isfinite(out).all()
    print("ccf=%s bs=%d %dx%d : %.2f ms/step  %.0f img/s  loss %.3f  eval ok" % (ccf, B, H, W, dt * 1e3, B / dt, lv), flush=True)
    del tr, model, base, x
    torch.cuda.empty_cache()
