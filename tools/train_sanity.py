"""Sanity: 300 optimizer steps on ONE fixed synthetic batch (bs 128, 224x224, 10 classes): the loss must fall steadily (the
network memorises the batch) and stay finite -- exercises fwd + bwd + Adam + BN running statistics end to end at scale."""
import sys, torch, contextlib, io
sys.path.insert(0, '.')
from mnasnet_pytorch_amd import FineTuneModelPool, load_model
from mnasnet_pytorch_amd.train_step import Trainer
dev = torch.device("cuda", 0)
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    base = load_model("mnasnet")
model = FineTuneModelPool(base, "mnasnet", 10, "512").to(dev).train()
tr = Trainer(model, lr=1e-3)
g = torch.Generator(device=dev).manual_seed(3)
x = torch.randn(128, 3, 224, 224, device=dev, generator=g)
y = torch.randint(0, 10, (128,), device=dev, generator=g)
losses = []
for i in range(300):
    losses.append(tr.step(x, y))
    if i % 50 == 49:
        print("step %3d loss %.4f" % (i + 1, float(losses[-1])), flush=True)
l = torch.stack(losses).float().cpu()
assert torch.isfinite(l).all()
assert l[-20:].mean() < 0.5 * l[:20].mean(), (float(l[:20].mean()), float(l[-20:].mean()))
with torch.no_grad():
    acc_t = (model(x).argmax(1) == y).float().mean()          # train-mode BatchNorm (batch statistics): memorised
model.eval()
with torch.no_grad():
    acc_e = (model(x).argmax(1) == y).float().mean()          # running statistics: chance level -- the fp32 CPU oracle of
print("first-20 mean loss %.3f, last-20 mean loss %.3f, accuracy on the batch: train-mode BN %.2f, eval-mode BN %.2f" % (
    l[:20].mean(), l[-20:].mean(), float(acc_t), float(acc_e)))  # the reference behaves the same on memorised noise
