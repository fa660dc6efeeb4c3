"""Native classifier head + loss: ``FineTuneModelPool.classifier`` (classifiers.py:56-89, an ``nn.Sequential`` of
Dropout / Linear / ReLU) and ``nn.CrossEntropyLoss`` (train.py:277) on the HIP library (csrc/mnas_head.hip).

Two entry points share the same kernels:

* :class:`NativeHead` ``.apply(x)`` -- an autograd function, used by ``FineTuneModelPool.forward`` so that
  ``model(x)`` / ``criterion(out, target)`` / ``loss.backward()`` of train.py:434-439 work unchanged;
* :meth:`NativeHead.loss_and_grad` -- logits, cross-entropy, and the whole backward of the head in 8 launches without
  autograd; ``Trainer.step`` takes this path when the criterion is a plain ``nn.CrossEntropyLoss``.

Dropout masks are a counter-based hash of (seed, element index) evaluated inside the GEMM kernels (never stored).  The seed
of a call is derived from ``torch.initial_seed()``, the rank and a per-head call counter, so runs are reproducible under
``torch.manual_seed`` but the mask stream is not ATen's Philox stream (the reference's masks differ from device to device
anyway).  No CPU fallback: tensors must live on the MI355X."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib as L

_MASK64 = (1 << 64) - 1


def _mix(seed: int, a: int, b: int) -> int:
    """64-bit seed of (call, layer): splitmix64 finaliser, same arithmetic as oracle.head_layer_seed"""
    z = (seed + 0x9E3779B97F4A7C15 * (a + 1) + 0xD1B54A32D192ED03 * (b + 1)) & _MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK64
    return z ^ (z >> 31)


class _Layer:
    __slots__ = ("p", "lin", "relu")

    def __init__(self, p, lin):
        self.p, self.lin, self.relu = p, lin, False


def parse_sequential(seq: nn.Module) -> Optional[List[_Layer]]:
    """[Dropout?] Linear [ReLU] ... -> layers, or None if the module is anything else (the caller then keeps PyTorch)."""
    if not isinstance(seq, nn.Sequential):
        return None
    layers: List[_Layer] = []
    pending = 0.0
    prev = None
    for m in seq:
        if isinstance(m, nn.Dropout):
            if prev == "drop" or not (0.0 <= m.p < 1.0):
                return None
            pending, prev = float(m.p), "drop"
        elif isinstance(m, nn.Linear):
            layers.append(_Layer(pending, m))
            pending, prev = 0.0, "lin"
        elif isinstance(m, nn.ReLU):
            if prev != "lin":
                return None
            layers[-1].relu, prev = True, "relu"
        else:
            return None
    if not layers or prev == "drop" or layers[-1].relu:
        return None            # trailing Dropout / ReLU on the logits: not one of the reference's heads
    for a, b in zip(layers, layers[1:]):
        if a.lin.out_features != b.lin.in_features:
            return None
    return layers


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, head, x, *params):
        seeds = head.next_seeds()
        us, logits = head.forward_layers(x, seeds)
        ctx.head, ctx.seeds, ctx.us = head, seeds, us
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        head = ctx.head
        grads = [(torch.empty_like(l.lin.weight), torch.empty_like(l.lin.bias) if l.lin.bias is not None else None)
                 for l in head.layers]
        dx = head.backward_layers(ctx.us, ctx.seeds, dlogits.contiguous().float(), grads, accumulate=False,
                                  need_dx=ctx.needs_input_grad[1])
        flat = []
        for (dw, db), l in zip(grads, head.layers):
            flat.append(dw)
            if l.lin.bias is not None:
                flat.append(db)
        return (None, dx) + tuple(flat)


class NativeHead:
    def __init__(self, layers: List[_Layer], owner: nn.Module):
        self.layers = layers
        self.owner = owner               # the nn.Sequential: .training decides whether dropout is applied
        self.lib = L.load()
        self.calls = 0
        rank = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank = torch.distributed.get_rank()
        self.seed0 = _mix(torch.initial_seed() & _MASK64, rank, 0x48454144)
        self._scratch = {}

    @staticmethod
    def build(seq: nn.Module) -> Optional["NativeHead"]:
        layers = parse_sequential(seq)
        return NativeHead(layers, seq) if layers is not None else None

    def params(self):
        out = []
        for l in self.layers:
            out.append(l.lin.weight)
            if l.lin.bias is not None:
                out.append(l.lin.bias)
        return out

    def next_seeds(self) -> List[int]:
        self.calls += 1
        return [_mix(self.seed0, self.calls, i) for i in range(len(self.layers))]

    # ---- kernels ----------------------------------------------------------------------------------------------------
    def _check(self, x):
        if x.device.type != "cuda":
            raise RuntimeError("the native head runs on the MI355X only (no CPU path)")
        if x.dim() != 2 or x.shape[1] != self.layers[0].lin.in_features:
            raise ValueError("head input must be (N, %d), got %s" % (self.layers[0].lin.in_features, tuple(x.shape)))

    def _desc(self, l: _Layer, N: int, seed: int) -> L.MnasHeadLinear:
        a = L.MnasHeadLinear()
        a.N, a.I, a.O = N, l.lin.in_features, l.lin.out_features
        a.relu = 1 if l.relu else 0
        a.drop_p = l.p if self.owner.training else 0.0
        a.seed = seed
        a.w = l.lin.weight.data_ptr()
        a.b = L.ptr(l.lin.bias)
        return a

    def forward_layers(self, x, seeds):
        """-> ([input of every layer], logits).  x: (N, I0) fp32 contiguous."""
        self._check(x)
        x = x.contiguous().float()
        N = x.shape[0]
        us = []
        for l, seed in zip(self.layers, seeds):
            if not (l.lin.weight.is_contiguous() and l.lin.weight.dtype == torch.float32):
                raise RuntimeError("head weights must be contiguous fp32")
            y = torch.empty((N, l.lin.out_features), dtype=torch.float32, device=x.device)
            a = self._desc(l, N, seed)
            a.x, a.y = x.data_ptr(), y.data_ptr()
            L.check(self.lib.mnas_head_linear_fwd(C.byref(a), L.cur_stream()), "head_linear_fwd")
            us.append(x)
            x = y
        return us, x

    def backward_layers(self, us, seeds, dz, grads, accumulate: bool, need_dx: bool = True):
        """dz: gradient of the logits.  grads[i] = (dW, db or None) tensors written (or accumulated into).  -> dx or None"""
        N = dz.shape[0]
        for i in range(len(self.layers) - 1, -1, -1):
            l = self.layers[i]
            a = self._desc(l, N, seeds[i])
            a.x, a.dz = us[i].data_ptr(), dz.data_ptr()
            a.dw, a.db = grads[i][0].data_ptr(), L.ptr(grads[i][1])
            a.accumulate = 1 if accumulate else 0
            L.check(self.lib.mnas_head_linear_bwd_w(C.byref(a), L.cur_stream()), "head_linear_bwd_w")
            if i == 0 and not need_dx:
                return None
            dx = torch.empty((N, l.lin.in_features), dtype=torch.float32, device=dz.device)
            a.dx = dx.data_ptr()
            a.relu_mask = us[i].data_ptr() if (i > 0 and self.layers[i - 1].relu) else None
            L.check(self.lib.mnas_head_linear_bwd_x(C.byref(a), L.cur_stream()), "head_linear_bwd_x")
            dz = dx
        return dz

    def cross_entropy(self, logits, target, ignore_index=-100, need_grad=True):
        """-> (loss 0-d tensor, dlogits or None).  nn.CrossEntropyLoss(reduction='mean') semantics."""
        N, Cn = logits.shape
        if target.dtype != torch.int64 or target.shape != (N,):
            raise ValueError("target must be int64 of shape (N,)")
        if target.device != logits.device:
            raise RuntimeError("target is on %s, logits on %s (the kernel takes raw device pointers)" % (target.device, logits.device))
        dev = logits.device
        key = (N, dev)
        if key not in self._scratch:
            self._scratch[key] = (torch.empty(N, dtype=torch.float32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))
        rows, bad = self._scratch[key]
        loss = torch.empty((), dtype=torch.float32, device=dev)
        dl = torch.empty_like(logits) if need_grad else None
        L.check(self.lib.mnas_head_cross_entropy(logits.data_ptr(), target.contiguous().data_ptr(), N, Cn, int(ignore_index),
                                                 rows.data_ptr(), loss.data_ptr(), L.ptr(dl), bad.data_ptr(), L.cur_stream()),
                "head_cross_entropy")
        return loss, dl

    def check_targets(self):
        """Host check of the device-side "target out of range" flag that mnas_head_cross_entropy raises (the loss is NaN from
        that step on; ATen asserts on the device instead).  Synchronises: call it next to a ``loss.item()``, not per step.
        Resets the flag."""
        for (N, dev), (_, bad) in self._scratch.items():
            if int(bad.item()) != 0:
                bad.zero_()
                raise ValueError("cross_entropy: a target index was outside [0, num_classes) and not ignore_index")

    # ---- entry points -----------------------------------------------------------------------------------------------
    def apply(self, x):
        return _HeadFn.apply(self, x, *self.params())

    def loss_and_grad(self, x, target, ignore_index=-100, need_dx=True):
        """Forward, loss and full backward of the head without autograd.  Parameter gradients are ACCUMULATED into
        ``p.grad`` (which must exist: Trainer points them into its flat gradient buffer).  -> (logits, loss, dx)"""
        seeds = self.next_seeds()
        us, logits = self.forward_layers(x, seeds)
        loss, dl = self.cross_entropy(logits, target, ignore_index)
        grads = []
        for l in self.layers:
            if l.lin.weight.grad is None or (l.lin.bias is not None and l.lin.bias.grad is None):
                raise RuntimeError("loss_and_grad needs preallocated .grad tensors on the head parameters")
            grads.append((l.lin.weight.grad, l.lin.bias.grad if l.lin.bias is not None else None))
        dx = self.backward_layers(us, seeds, dl, grads, accumulate=True, need_dx=need_dx)
        return logits, loss, dx

    def dropout_mask(self, layer: int, N: int, seed: int):
        """keep mask (N, in_features) uint8 of one layer for a given seed (tests)."""
        l = self.layers[layer]
        out = torch.empty((N, l.lin.in_features), dtype=torch.uint8, device="cuda")
        L.check(self.lib.mnas_head_dropout_mask(out.data_ptr(), out.numel(), l.p, seed, L.cur_stream()), "head_dropout_mask")
        return out
