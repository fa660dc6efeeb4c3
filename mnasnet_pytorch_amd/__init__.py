"""mnasnet_pytorch_amd -- MI355X (gfx950) native MNASNet training hot path.

Drop-in for snakers4/mnasnet-pytorch's src/models/mnasnet.py nn.Module surface and src/train.py's step
loop.  Host code is PyTorch-ROCm (device memory, streams, torch.distributed/RCCL); all conv / BatchNorm /
ReLU arithmetic runs in hand-written HIP kernels behind the C ABI in include/mnas.h.
(The directory is `mnasnet_pytorch_amd` -- a hyphen is not importable in Python.)
"""
from . import _lib  # noqa: F401
