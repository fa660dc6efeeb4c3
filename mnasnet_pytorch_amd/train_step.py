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


class Trainer:
    """Owns flat parameter / gradient / Adam-moment buffers for ``model`` (a FineTuneModelPool or anything with
    a ``features`` engine module plus ordinary PyTorch head parameters) and runs train.py's step."""

    def __init__(self, model: nn.Module, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 criterion: Optional[nn.Module] = None, distributed: bool = False, process_group=None,
                 early_bucket_stage: int = 5):
        self.model = model
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.criterion = criterion if criterion is not None else nn.CrossEntropyLoss()   # train.py:277
        self.step_count = 0
        self.lib = L.load()
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("Trainer needs the model on an MI355X device (no CPU path)")
        self.device = dev
        feats = model.features if hasattr(model, "features") else model
        self.engine = feats._engine()
        eng_params = list(self.engine.params)
        eng_ids = {id(p) for p in eng_params}
        head = [p for p in model.parameters() if id(p) not in eng_ids and p.requires_grad]
        self.head_params = head
        n_head = sum(p.numel() for p in head)
        n_eng = self.engine.grad_numel
        n = n_head + n_eng
        self.flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
        # ---- parameters become views of flat_p (same element order as the gradient layout)
        off = 0
        with torch.no_grad():
            for p in head:
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat_p[off:off + k].view(p.shape)
                p.grad = self.flat_g[off:off + k].view(p.shape)      # autograd accumulates in place
                off += k
            for ci in sorted(self.engine.convs, key=lambda c: -c.stage):
                for j, p in enumerate(ci.params):
                    o, k = ci.gslice[j]
                    self.flat_p[n_head + o:n_head + o + k].copy_(p.reshape(-1))
                    p.data = self.flat_p[n_head + o:n_head + o + k].view(p.shape)
        self.engine.bind_grad_buffer(self.flat_g[n_head:])
        self.engine.ensure_setup(dev)
        for p, v in zip(self.engine.params, self.engine.grad_views):
            if p.requires_grad:
                p.grad = v                                            # "already ours" -> engine accumulates
        # ---- data parallel
        self.distributed = distributed
        self.world = 1
        self.buckets = None
        if distributed:
            import torch.distributed as dist
            self.world = dist.get_world_size(process_group)
            dist.broadcast(self.flat_p, src=0, group=process_group)     # identical replicas (DP semantics)
            # bucket 0 ends where stage `early_bucket_stage`'s gradients end
            rng = self.engine.stage_ranges
            stages_early = [s for s in rng if s >= early_bucket_stage]
            split = n_head + (max(rng[s][1] for s in stages_early) if stages_early else 0)
            self.buckets = FlatBuckets(self.flat_g, [0, split, n], process_group)
            self._early_stage = min(stages_early) if stages_early else None
            self.engine.on_stage_done = self._on_stage_done
        self._launched0 = False

    # engine callback: backward of features.<stage> has been enqueued
    def _on_stage_done(self, stage: int):
        if self.buckets is not None and not self._launched0 and self._early_stage is not None and stage <= self._early_stage:
            self.buckets.launch(0)
            self._launched0 = True

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
        self.flat_g.zero_()                                  # optimizer.zero_grad()
        self._launched0 = False
        out = self.model(x.float())
        loss = self.criterion(out, target)
        loss.backward()
        if self.buckets is not None:
            if not self._launched0:
                self.buckets.launch(0)
            self.buckets.launch(1)
            self.buckets.wait()
        self.step_count += 1
        L.check(self.lib.mnas_adam_step(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(),
                                        self.flat_v.data_ptr(), self.flat_p.numel(), self.lr, self.betas[0],
                                        self.betas[1], self.eps, self.wd, self.step_count, 1.0 / self.world,
                                        L.cur_stream()), "adam_step")
        return loss.detach()
