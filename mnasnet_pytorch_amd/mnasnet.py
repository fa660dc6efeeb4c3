"""Drop-in for the reference's ``src/models/mnasnet.py`` nn.Module surface, executed by the HIP engine.

Same class names, constructor signatures, attribute names and ``state_dict`` keys as the reference
(/root/reference/src/models/mnasnet.py:37-213): ``ConvBlock``, ``SepConv``, ``MBConv_block``, ``MBConv``,
``Mnasnet(cut_channels_first=True)``, ``.features`` (an ``nn.Sequential`` of 8), ``.init_params()``.
The ``nn.Conv2d`` / ``nn.BatchNorm2d`` children are kept as PARAMETER CONTAINERS only (so ``state_dict``,
``load_state_dict``, ``.to()``, optimizers and checkpoints behave exactly as with the reference); their own
``forward`` is never called.  Every module's ``forward`` hands its whole subtree to
``engine.Engine``, which runs the hand-written gfx950 kernels (include/mnas.h).  There is no eager /
CPU fallback: calling ``forward`` on a CPU tensor raises.

Reference quirks kept on purpose: list-multiplied blocks share ONE module object
(weights, BN affine and running statistics; mnasnet.py:162-164), conv bias before BN (:54), ReLU after the
projection and before the residual add (:126-133), the extra dense 3x3 per stage (:157-161), the ignored
``momentum`` ctor argument (:46,55).
"""
import torch
import torch.nn as nn
from torch.nn import init

from .engine import Engine

default_activation = nn.ReLU   # mnasnet.py:9

__all__ = ["ConvBlock", "SepConv", "MBConv_block", "MBConv", "Mnasnet", "SqueezeExcite"]


class _EngineModule(nn.Module):
    """Mixin: forward() = run this module's subtree on the HIP engine (built lazily, cached per shape)."""

    def _engine(self) -> Engine:
        eng = self.__dict__.get("_mnas_engine")
        if eng is None:
            eng = Engine(self)
            self.__dict__["_mnas_engine"] = eng      # not a submodule / not in state_dict / not pickled
        return eng

    def forward(self, x):
        return self._engine().forward(x)

    def __getstate__(self):
        d = self.__dict__.copy()
        d.pop("_mnas_engine", None)
        return d


class ConvBlock(_EngineModule):
    """relu(bn(conv(x)))  -- mnasnet.py:37-62"""

    def __init__(self, in_, out_, kernel_size=3, stride=1, padding=0, groups=1, activation=default_activation,
                 momentum=0.1):
        super().__init__()
        if activation is not nn.ReLU:
            raise NotImplementedError("the HIP path fuses ReLU (the reference's default_activation)")
        self.conv = nn.Conv2d(in_, out_, kernel_size=kernel_size, stride=stride, padding=padding, groups=groups,
                              bias=True)
        self.bn = nn.BatchNorm2d(out_)           # `momentum` is NOT forwarded, as in the reference (:55)
        self.activation = activation(inplace=True)


class SepConv(_EngineModule):
    """depthwise kxk + pointwise 1x1, no skip  -- mnasnet.py:64-103"""

    def __init__(self, in_channels, out_channels, kernel_size=3, reduce=False, repeat=0):
        super().__init__()
        padding = kernel_size // 2
        stride = 2 if reduce else 1
        seq = [ConvBlock(in_channels, in_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                         groups=in_channels),
               ConvBlock(in_channels, in_channels, kernel_size=1, stride=1)] * repeat + \
              [ConvBlock(in_channels, in_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                         groups=in_channels),
               ConvBlock(in_channels, out_channels, kernel_size=1, stride=1)]
        self.sequence = nn.Sequential(*seq)


class SqueezeExcite(nn.Module):
    """Squeeze-and-excitation on the activated depthwise output of an MBConv_block: a * sigmoid(fc2(relu(fc1(mean_hw a)))).
    BUILD-DEFINED (BASELINE config 4's "SE-block variant"): the reference has no SE block; the definition is restated in
    oracle.mnasnet_oracle.se_apply and that is what the parity tests check ("parity unpinned by the reference").  A parameter
    container like ConvBlock's children: the HIP engine runs it (csrc/mnas_se.hip); its own forward is never called."""

    def __init__(self, channels, reduced):
        super().__init__()
        self.fc1 = nn.Linear(channels, reduced)
        self.fc2 = nn.Linear(reduced, channels)


class MBConv_block(_EngineModule):
    """x + project(dw(expand(x)))  -- the inverted-residual block, mnasnet.py:105-137.
    se_ratio > 0 (not in the reference): a SqueezeExcite with max(8, in_channels * se_ratio) hidden units between the depthwise
    and the projection ConvBlocks (extra state_dict keys ``se.fc1.*`` / ``se.fc2.*``; with the default 0 the key set is the
    reference's)."""

    def __init__(self, in_channels, channel_factor, kernel_size=3, se_ratio=0.0):
        super().__init__()
        self.in_channels = in_channels
        padding = kernel_size // 2
        mid = in_channels * channel_factor
        self.sequence = nn.Sequential(
            ConvBlock(in_channels, mid, kernel_size=1, stride=1),
            ConvBlock(mid, mid, kernel_size=kernel_size, stride=1, padding=padding, groups=mid),
            ConvBlock(mid, in_channels, kernel_size=1, stride=1))
        if se_ratio and se_ratio > 0:
            self.se = SqueezeExcite(mid, max(8, int(round(in_channels * se_ratio))))


class MBConv(_EngineModule):
    """stage container  -- mnasnet.py:139-173 (list-multiply => ONE shared block applied `layers` times)"""

    def __init__(self, in_channels, out_channels, channel_factor, layers, kernel_size=3, reduce=True,
                 cut_channels_first=True, se_ratio=0.0):
        super().__init__()
        block_channels = out_channels if cut_channels_first else in_channels
        stride = 2 if reduce else 1
        seq = [ConvBlock(in_channels, out_channels, kernel_size=3, stride=stride, padding=1)] + \
              [MBConv_block(block_channels, channel_factor, kernel_size, se_ratio=se_ratio)] * layers
        if not cut_channels_first:
            seq = list(reversed(seq))
        self.sequence = nn.Sequential(*seq)


class _Features(nn.Sequential):
    """``Mnasnet.features``: an nn.Sequential (same child names '0'..'7') whose forward is ONE engine
    program over all eight children instead of eight separate ones.  classifiers.py:47 re-parents it."""

    def _engine(self) -> Engine:
        eng = self.__dict__.get("_mnas_engine")
        if eng is None:
            eng = Engine(self)
            self.__dict__["_mnas_engine"] = eng
        return eng

    def forward(self, x):
        return self._engine().forward(x)

    def __getstate__(self):
        d = self.__dict__.copy()
        d.pop("_mnas_engine", None)
        return d


class Mnasnet(nn.Module):
    """mnasnet.py:175-213.  ``Mnasnet(cut_channels_first=True)`` is the reference's signature and network.  The keyword-only
    extras build BASELINE config 4's variant (not in the reference): ``kernel_size=5`` gives every MBConv stage 5x5 depthwise
    convs, ``se_ratio=0.25`` adds a SqueezeExcite to every MBConv_block."""

    def __init__(self, cut_channels_first=True, *, kernel_size=None, se_ratio=0.0):
        super().__init__()
        ccf = cut_channels_first
        ks = (lambda k: k) if kernel_size is None else (lambda k: kernel_size)
        kw = dict(cut_channels_first=ccf, se_ratio=se_ratio)
        self.features = _Features(
            ConvBlock(3, 32, kernel_size=3, stride=2, padding=1),
            SepConv(32, 16, kernel_size=3),
            MBConv(16, 24, channel_factor=3, layers=3, kernel_size=ks(3), reduce=True, **kw),
            MBConv(24, 40, channel_factor=3, layers=3, kernel_size=ks(5), reduce=True, **kw),
            MBConv(40, 80, channel_factor=6, layers=3, kernel_size=ks(5), reduce=True, **kw),
            MBConv(80, 96, channel_factor=6, layers=2, kernel_size=ks(3), reduce=False, **kw),
            MBConv(96, 192, channel_factor=6, layers=4, kernel_size=ks(5), reduce=True, **kw),
            MBConv(192, 320, channel_factor=6, layers=1, kernel_size=ks(3), reduce=False, **kw))
        self.init_params()

    def init_params(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                init.kaiming_normal_(m.weight, mode="fan_out")
                if m.bias is not None:
                    init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                init.constant_(m.weight, 1)
                init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                init.normal_(m.weight, std=0.001)
                if m.bias is not None:
                    init.constant_(m.bias, 0)

    def forward(self, input):
        return self.features(input)
