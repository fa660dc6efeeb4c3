"""mnasnet_pytorch_amd -- MI355X (gfx950) native MNASNet training hot path.

Drop-in for snakers4/mnasnet-pytorch's src/models/mnasnet.py nn.Module surface and src/train.py's step
loop.  Host code is PyTorch-ROCm (device memory, streams, torch.distributed/RCCL); all conv / BatchNorm /
ReLU arithmetic runs in hand-written HIP kernels behind the C ABI in include/mnas.h.
(The directory is `mnasnet_pytorch_amd` -- a hyphen is not importable in Python.)
"""
from . import _lib  # noqa: F401
from .mnasnet import ConvBlock, MBConv, MBConv_block, Mnasnet, SepConv  # noqa: F401
from .classifiers import FineTuneModelPool, load_model  # noqa: F401
from .sampler import ClusterRandomSampler, DistributedClusterSampler  # noqa: F401

__all__ = ["Mnasnet", "ConvBlock", "SepConv", "MBConv_block", "MBConv", "load_model", "FineTuneModelPool",
           "ClusterRandomSampler", "DistributedClusterSampler"]
