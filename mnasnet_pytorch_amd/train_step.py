"""The training step of the reference (``src/train.py:423-440``) on MI355X, single- and multi-GPU.

    input = input.float().to(device); out = model(input); loss = criterion(out, target)
    optimizer.zero_grad(); loss.backward(); optimizer.step()

Reference parallelism: one ``torch.nn.DataParallel`` call (train.py:202): single process, batch scattered
along dim 0, per-replica (UNSYNCED) BatchNorm statistics, gradients reduce-added, replica-0 buffers win.
Here: one process per GPU (``torch.distributed`` backend "nccl" = RCCL over xGMI), the same per-rank BatchNorm
semantics, and ONE flat fp32 gradient buffer laid out in the order gradients become final during backward
    [ classifier head | features.7 | features.6 | ... | features.0 ]
so the all-reduce runs as two contiguous buckets: bucket 0 (head + late stages: >80 % of the bytes) is
launched from the engine's stage-done callback while the early, activation-heavy stages are still running
backward; bucket 1 at the end.  The payload is ~8.9 MB: latency-bound on xGMI, so few large buckets, not many
small ones.  The optimizer is one fused Adam launch over the flat parameter buffer
(``mnas_adam_step``; torch.optim.Adam semantics, train.py:219-221), with the 1/world_size averaging folded in.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib as L


class FlatBuckets:
    """Contiguous [start, end) element ranges of a flat gradient buffer, all-reduced asynchronously.
    Pure host logic + torch.distributed: exercised on CPU/gloo by tests/test_ddp_gloo.py."""

    def __init__(self, flat: torch.Tensor, bounds: List[int], group=None):
        assert bounds[0] == 0 and bounds[-1] == flat.numel() and all(a <= b for a, b in zip(bounds, bounds[1:]))
        self.flat, self.bounds, self.group = flat, bounds, group
        self.handles = []

    @property
    def n(self):
        return len(self.bounds) - 1

    def launch(self, i: int):
        import torch.distributed as dist
        a, b = self.bounds[i], self.bounds[i + 1]
        if b > a:
            self.handles.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        for h in self.handles:
            h.wait()
        self.handles = []


class BucketSchedule:
    """WHEN the two gradient buckets are all-reduced during backward (pure host logic; tests/test_ddp_gloo.py drives it
    with a stub engine on CPU/gloo).

    The flat gradient buffer is ``[head | engine stages, later stages first]``.  Bucket 0 = head + every stage >=
    ``early_bucket_stage`` (>80 % of the bytes: the late stages hold the wide weights) and is complete as soon as the backward
    of stage ``early_stage`` (= the smallest such stage present) has been enqueued -- the engine's stage-done callback fires
    right there, so the all-reduce runs under the backward of the early, activation-heavy stages.  Bucket 1 = the rest, launched
    at the end of backward.  ``join_stages`` = the stages after which the engine must join its weight-gradient side stream
    (bucket 0's gradients have to be complete on the stream the collective is ordered after)."""

    def __init__(self, flat_g: torch.Tensor, n_head: int, stage_ranges, early_bucket_stage: int, group=None):
        n = flat_g.numel()
        early = [s for s in stage_ranges if s >= early_bucket_stage]
        split = n_head + (max(stage_ranges[s][1] for s in early) if early else 0)
        self.buckets = FlatBuckets(flat_g, [0, split, n], group)
        self.early_stage = min(early) if early else None
        self.join_stages = {self.early_stage} if self.early_stage is not None else set()
        self.launched0 = False
        self.log = []                     # ("stage", s) / ("launch", bucket): the order things happened in (tests)

    def begin_step(self):
        self.launched0 = False
        self.log = []

    def on_stage_done(self, stage: int):
        """engine callback: the backward of features.<stage> (and everything after it) has been enqueued"""
        self.log.append(("stage", stage))
        if not self.launched0 and self.early_stage is not None and stage <= self.early_stage:
            self.buckets.launch(0)
            self.launched0 = True
            self.log.append(("launch", 0))

    def finish(self):
        """end of backward: whatever has not been launched yet, then wait for both"""
        if not self.launched0:
            self.buckets.launch(0)
            self.launched0 = True
            self.log.append(("launch", 0))
        self.buckets.launch(1)
        self.log.append(("launch", 1))
        self.buckets.wait()


class FlatAdam(torch.optim.Optimizer):
    """``torch.optim.Adam`` (train.py:219-221) over ONE flat fp32 buffer: a single fused launch (``mnas_adam_step``) instead
    of ~110 per-tensor updates.  It is a real ``torch.optim.Optimizer``:

    * ``param_groups[0]['lr']`` (and betas / eps / weight_decay) are read at every step, so the schedulers of the
      reference (MultiStepLR, ExponentialLR, ReduceLROnPlateau, CyclicLR.batch_step: train.py:284-337) and the
      ``optimizer.state_dict()['param_groups'][0]['lr']`` read-back (train.py:450) work unchanged;
    * ``state_dict()`` / ``load_state_dict()`` carry the moments and the step count (train.py:382 checkpoints them);
    * parameters with ``requires_grad == False`` (``FineTuneModelPool.freeze()``, classifiers.py:95-99) are left
      untouched: the update runs over the contiguous runs of trainable elements only, so weight decay cannot move a
      frozen weight.
    """

    def __init__(self, params, flat_p, flat_g, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
        params = list(params)
        n = sum(p.numel() for p in params)
        if flat_p.numel() != n or flat_g.numel() != n:
            raise ValueError("flat buffers must hold exactly the parameters handed to FlatAdam")
        off = 0
        self._ranges = []                 # element range of every parameter inside the flat buffers
        for p in params:
            if p.data_ptr() != flat_p.data_ptr() + 4 * off:
                raise ValueError("parameters must be views of flat_p in order")
            self._ranges.append((off, off + p.numel()))
            off += p.numel()
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.flat_p, self.flat_g = flat_p, flat_g
        self.flat_m = torch.zeros_like(flat_p)
        self.flat_v = torch.zeros_like(flat_p)
        self.step_count = 0
        self.grad_scale = grad_scale      # 1/world_size: the all-reduce sums, Adam sees the mean (DataParallel semantics)
        self._lib = L.load()
        self._runs_key, self._runs = None, []

    def _trainable_runs(self):
        ps = self.param_groups[0]["params"]
        key = tuple(p.requires_grad for p in ps)
        if key != self._runs_key:
            runs = []
            for (a, b), rg in zip(self._ranges, key):
                if not rg or a == b:
                    continue
                if runs and runs[-1][1] == a:
                    runs[-1][1] = b
                else:
                    runs.append([a, b])
            self._runs_key, self._runs = key, runs
        return self._runs

    def zero_grad(self, set_to_none: bool = False):
        """The gradients are views of the flat buffer: zero it in one launch (never set to None)."""
        self.flat_g.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]
        self.step_count += 1
        for a, b in self._trainable_runs():
            self._launch(a, b, g)
        return loss

    def _launch(self, a, b, g):
        L.check(self._lib.mnas_adam_step(self.flat_p.data_ptr() + 4 * a, self.flat_g.data_ptr() + 4 * a,
                                         self.flat_m.data_ptr() + 4 * a, self.flat_v.data_ptr() + 4 * a, b - a,
                                         float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                         float(g["weight_decay"]), self.step_count, float(self.grad_scale),
                                         L.cur_stream()), "adam_step")

    def state_dict(self):
        g = self.param_groups[0]
        return {"state": {"step": self.step_count, "exp_avg": self.flat_m.clone(), "exp_avg_sq": self.flat_v.clone()},
                "param_groups": [{k: v for k, v in g.items() if k != "params"} | {"params": list(range(len(g["params"])))}]}

    def load_state_dict(self, sd):
        st = sd["state"]
        if st["exp_avg"].numel() != self.flat_m.numel():
            raise ValueError("optimizer state has %d elements, this model %d" % (st["exp_avg"].numel(), self.flat_m.numel()))
        self.step_count = int(st["step"])
        self.flat_m.copy_(st["exp_avg"].to(self.flat_m.device))
        self.flat_v.copy_(st["exp_avg_sq"].to(self.flat_v.device))
        for k, v in sd["param_groups"][0].items():
            if k != "params":
                self.param_groups[0][k] = v


class FlatRMSprop(FlatAdam):
    """``torch.optim.RMSprop`` (train.py:222-224: ``--optimizer rmsprop``; alpha 0.99, eps 1e-8, centered = False) over the flat
    buffers: one fused launch (``mnas_rmsprop_step``).  Same Optimizer contract as :class:`FlatAdam`."""

    def __init__(self, params, flat_p, flat_g, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0.0, momentum=0.0, centered=False,
                 grad_scale=1.0):
        if centered:
            raise NotImplementedError("centered RMSprop is not fused (the reference never asks for it: train.py:222-224)")
        super().__init__(params, flat_p, flat_g, lr=lr, eps=eps, weight_decay=weight_decay, grad_scale=grad_scale)
        self.param_groups[0].pop("betas", None)
        self.param_groups[0].update(alpha=alpha, momentum=momentum, centered=False)
        # flat_m = momentum buffer, flat_v = square average (same checkpoint layout as FlatAdam: exp_avg / exp_avg_sq)

    def _launch(self, a, b, g):
        L.check(self._lib.mnas_rmsprop_step(self.flat_p.data_ptr() + 4 * a, self.flat_g.data_ptr() + 4 * a,
                                            self.flat_v.data_ptr() + 4 * a, self.flat_m.data_ptr() + 4 * a, b - a, float(g["lr"]),
                                            float(g["alpha"]), float(g["eps"]), float(g["weight_decay"]), float(g["momentum"]),
                                            float(self.grad_scale), L.cur_stream()), "rmsprop_step")


class FlatSGD(FlatAdam):
    """``torch.optim.SGD`` (train.py:226-228: ``--optimizer sgd``; momentum 0 by default) over the flat buffers
    (``mnas_sgd_step``).  ``flat_m`` is the momentum buffer (initialised with the first gradient, as torch does)."""

    def __init__(self, params, flat_p, flat_g, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, grad_scale=1.0):
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        super().__init__(params, flat_p, flat_g, lr=lr, weight_decay=weight_decay, grad_scale=grad_scale)
        for k in ("betas", "eps"):
            self.param_groups[0].pop(k, None)
        self.param_groups[0].update(momentum=momentum, dampening=dampening, nesterov=nesterov)

    def _launch(self, a, b, g):
        L.check(self._lib.mnas_sgd_step(self.flat_p.data_ptr() + 4 * a, self.flat_g.data_ptr() + 4 * a,
                                        self.flat_m.data_ptr() + 4 * a, b - a, float(g["lr"]), float(g["momentum"]),
                                        float(g["dampening"]), float(g["weight_decay"]), 1 if g["nesterov"] else 0, self.step_count,
                                        float(self.grad_scale), L.cur_stream()), "sgd_step")


class Trainer:
    """Owns flat parameter / gradient / Adam-moment buffers for ``model`` (a FineTuneModelPool or anything with
    a ``features`` engine module plus ordinary PyTorch head parameters) and runs train.py's step.
    ``trainer.optimizer`` is a :class:`FlatAdam` (a ``torch.optim.Optimizer``): hand it to the reference's schedulers and
    checkpoint it with ``optimizer.state_dict()`` exactly as train.py does."""

    def __init__(self, model: nn.Module, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 criterion: Optional[nn.Module] = None, distributed: bool = False, process_group=None,
                 early_bucket_stage: int = 5, optimizer: str = "adam", **optimizer_kwargs):
        self.model = model
        self.native_step = True          # see _native_head()
        self.last_logits = None
        self.criterion = criterion if criterion is not None else nn.CrossEntropyLoss()   # train.py:277
        self.lib = L.load()
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("Trainer needs the model on an MI355X device (no CPU path)")
        self.device = dev
        feats = model.features if hasattr(model, "features") else model
        self.engine = feats._engine()
        eng_params = list(self.engine.params)
        eng_ids = {id(p) for p in eng_params}
        head = [p for p in model.parameters() if id(p) not in eng_ids]
        self.head_params = head
        n_head = sum(p.numel() for p in head)
        n_eng = self.engine.grad_numel
        n = n_head + n_eng
        self.flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        # ---- parameters become views of flat_p (same element order as the gradient layout)
        off = 0
        with torch.no_grad():
            for p in head:
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat_p[off:off + k].view(p.shape)
                p.grad = self.flat_g[off:off + k].view(p.shape)      # autograd accumulates in place
                off += k
            for ci in sorted(self.engine.param_owners, key=lambda c: -c.stage):
                for j, p in enumerate(ci.params):
                    o, k = ci.gslice[j]
                    self.flat_p[n_head + o:n_head + o + k].copy_(p.reshape(-1))
                    p.data = self.flat_p[n_head + o:n_head + o + k].view(p.shape)
        self.engine.bind_grad_buffer(self.flat_g[n_head:])
        self.engine.ensure_setup(dev)
        for p, v in zip(self.engine.params, self.engine.grad_views):
            p.grad = v                                                # "already ours" -> engine accumulates
        # flat order = head parameters, then the engine's (later stages first): FlatAdam checks the views line up
        # ---- data parallel
        self.distributed = distributed
        self.world = 1
        self.schedule: Optional[BucketSchedule] = None
        if distributed:
            import torch.distributed as dist
            self.world = dist.get_world_size(process_group)
            dist.broadcast(self.flat_p, src=0, group=process_group)     # identical replicas (DP semantics)
            self.schedule = BucketSchedule(self.flat_g, n_head, self.engine.stage_ranges, early_bucket_stage, process_group)
            self.engine.on_stage_done = self.schedule.on_stage_done
            # bucket 0 is all-reduced as soon as its last stage is done: its side-stream weight gradients must be in
            self.engine.join_stages = self.schedule.join_stages
            self.engine.reset_programs()
        # train.py:218-231: --optimizer adam | rmsprop | sgd, each constructed with lr only
        if optimizer.startswith("adam"):
            self.optimizer = FlatAdam(head + eng_params, self.flat_p, self.flat_g, lr=lr, betas=betas, eps=eps,
                                      weight_decay=weight_decay, grad_scale=1.0 / self.world, **optimizer_kwargs)
        elif optimizer.startswith("rmsprop"):
            self.optimizer = FlatRMSprop(head + eng_params, self.flat_p, self.flat_g, lr=lr, weight_decay=weight_decay,
                                         grad_scale=1.0 / self.world, **optimizer_kwargs)
        elif optimizer.startswith("sgd"):
            self.optimizer = FlatSGD(head + eng_params, self.flat_p, self.flat_g, lr=lr, weight_decay=weight_decay,
                                     grad_scale=1.0 / self.world, **optimizer_kwargs)
        else:
            raise ValueError("Optimizer not supported")            # train.py:231

    # convenience mirrors of the optimizer's hyper-parameters / state
    @property
    def lr(self):
        return self.optimizer.param_groups[0]["lr"]

    @lr.setter
    def lr(self, v):
        self.optimizer.param_groups[0]["lr"] = v

    @property
    def step_count(self):
        return self.optimizer.step_count

    @property
    def flat_m(self):
        return self.optimizer.flat_m

    @property
    def flat_v(self):
        return self.optimizer.flat_v

    def state_dict(self):
        """{'optimizer': FlatAdam.state_dict()} -- what train.py:382 stores under 'optimizer'."""
        return {"optimizer": self.optimizer.state_dict()}

    def load_state_dict(self, sd):
        self.optimizer.load_state_dict(sd["optimizer"])

    def _native_head(self):
        """The model's NativeHead when the whole step can bypass autograd: FineTuneModelPool-like model in training mode
        (fused pool, Dropout/Linear/ReLU classifier) and a plain mean-reduced nn.CrossEntropyLoss (train.py:277)."""
        m, c = self.model, self.criterion
        if not self.native_step or type(c) is not nn.CrossEntropyLoss:
            return None
        if c.weight is not None or c.reduction != "mean" or getattr(c, "label_smoothing", 0.0) != 0.0:
            return None
        if not (hasattr(m, "_native_head") and getattr(m, "native_head", False) and getattr(m, "fuse_pool", False)):
            return None
        if not (m.training and self.engine.root.training and m._pool_is_global_average()):
            return None
        if self.engine.root is not m.features or any(not p.requires_grad for p in self.head_params):
            return None
        if any(not p.requires_grad for p in self.engine.params):
            return None            # frozen features (FineTuneModelPool.freeze()): the module path skips their backward
        return m._native_head()

    @property
    def buckets(self):
        return self.schedule.buckets if self.schedule is not None else None

    def sync_buffers(self):
        """Rank-0 BatchNorm running statistics win (DataParallel semantics, train.py:202) -- call before
        validation / checkpointing."""
        if not self.distributed:
            return
        import torch.distributed as dist
        for b in self.model.buffers():
            dist.broadcast(b, src=0, group=self.buckets.group)

    def step(self, x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """One iteration of train.py:427-440.  Returns the loss tensor (no host sync)."""
        self.optimizer.zero_grad()                           # train.py:438
        if self.schedule is not None:
            self.schedule.begin_step()
        loss = self.forward_backward(x, target)
        if self.schedule is not None:
            self.schedule.finish()
        self.optimizer.step()                                # train.py:440
        return loss

    def forward_backward(self, x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """train.py:427-439 without the optimizer: forward, loss, backward INTO the flat gradient buffer (accumulating into what is
        already there: two calls between ``optimizer.zero_grad()`` and ``optimizer.step()`` sum their gradients, which is what
        the world-2 test uses to emulate two data-parallel ranks in one process).  Fires the engine's stage-done callback."""
        head = self._native_head()
        if head is not None:
            # features -> pool -> head -> cross-entropy -> head backward -> features backward as plain launch lists: no
            # autograd graph, no ATen kernels (csrc/mnas_head.hip); same arithmetic as the module path below
            eng = self.engine
            if hasattr(self.model, "_sync_input_norm"):
                self.model._sync_input_norm()
            eng.check_input(x)                               # same checks as Engine.forward: the launch lists take raw pointers
            if not isinstance(target, torch.Tensor) or target.device != x.device:
                raise RuntimeError("target must be a tensor on the input's device (%s)" % (x.device,))
            u8 = x.dtype == torch.uint8 and eng._in_norm is not None
            x = x.contiguous() if u8 else x.float().contiguous()
            eng.ensure_setup(x.device)
            eng._check_modes()
            prog = eng.program(x.shape[0], x.shape[2], x.shape[3], True, False, True, u8)
            f = prog.run_forward(x, static_io=True)
            head.calls = self.optimizer.step_count           # dropout masks follow the CHECKPOINTED step count: a resumed run does
                                                             # not replay the masks of the first steps
            self.last_logits, loss, df = head.loss_and_grad(f.view(f.size(0), -1), target, self.criterion.ignore_index)
            accumulate = eng.prepare_grads()
            prog.run_backward(df, eng.on_stage_done, static_io=True)
            eng.finish_grads(accumulate)
        else:
            out = self.model(x.float())
            loss = self.criterion(out, target)
            loss.backward()
        return loss.detach()
