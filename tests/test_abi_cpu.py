"""CPU (no GPU needed): the C-ABI library builds for gfx950, loads, and exports every symbol include/mnas.h
declares; the ctypes struct mirrors have the C sizes; product modules construct on CPU with the reference's
state_dict surface and refuse to run without an MI355X (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

import cases as C
from cases import O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from mnasnet_pytorch_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "mnas.h")).read()
    declared = set(re.findall(r"\b(mnas_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), name
        assert name in _lib.SYMBOLS, "ctypes prototype missing for " + name
    assert lib.mnas_version() == _lib.ABI_VERSION == 8 and lib.mnas_arch() == b"gfx950"
    assert ctypes.sizeof(_lib.MnasOp) == 4 + 15 * 4 + 4 * 8 + 16 * 8
    assert lib.mnas_packed_bytes(_lib.PACK_FWD, 48, 16, 1, 1) == 48 * 32 * 2
    assert lib.mnas_packed_bytes(_lib.PACK_DGRAD, 48, 16, 1, 1) == 16 * 64 * 2
    assert lib.mnas_packed_bytes(_lib.PACK_DW, 72, 1, 5, 5) == 25 * 72 * 4
    assert lib.mnas_workspace_bytes(0, 1024, 48, 0) == 2 * 48 * 1024 * 4        # conv statistics table
    assert lib.mnas_workspace_bytes(2, 512, 240, 40) == 512 * 240 * 40 * 4      # fused 1x1 backward slabs
    assert lib.mnas_workspace_bytes(9, 1, 1, 1) == -1
    assert lib.mnas_pw_bwd_supported(48, 16) == 1 and lib.mnas_pw_bwd_supported(96, 576) == 0   # host-side query
    assert ctypes.sizeof(_lib.MnasPackDesc) == 32


def test_opcode_constants_match_header():
    from mnasnet_pytorch_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mnas.h")).read()
    for name, val in re.findall(r"#define MNAS_OP_([A-Z_]+)\s+(\d+)", hdr):
        py = {"NCHW_TO_NHWC": "OP_NCHW_TO_NHWC"}.get(name, "OP_" + name)
        assert getattr(_lib, py) == int(val), name


@pytest.mark.parametrize("ccf", [True, False])
def test_module_surface_matches_reference(ccf):
    from mnasnet_pytorch_amd import Mnasnet
    m = Mnasnet(cut_channels_first=ccf)
    sd = m.state_dict()
    assert list(sd.keys()) == O.state_keys(ccf) and len(sd) == 399
    ref = O.init_state(ccf, C.STATE_SEED)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(ref[k].shape) and v.dtype == ref[k].dtype, k
    assert sum(p.numel() for p in m.parameters()) == (2799728 if ccf else 1541048)
    m.load_state_dict(ref)                                                                # checkpoints load
    # list-multiplied blocks share ONE module (mnasnet.py:162-164)
    seq = m.features[2].sequence
    blocks = [b for b in seq if type(b).__name__ == "MBConv_block"]
    assert len(blocks) == 3 and blocks[0] is blocks[1] is blocks[2]
    assert isinstance(m.features, torch.nn.Sequential) and len(m.features) == 8
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 32, 32))           # CPU tensor: no fallback


def test_finetune_pool_surface():
    from mnasnet_pytorch_amd import FineTuneModelPool, load_model
    base = load_model("mnasnet")
    for cfg in C.HEADS:
        m = FineTuneModelPool(base, "mnasnet", 10, cfg)
        assert m.features is base.features
        keys = [k for k in m.state_dict() if k.startswith("classifier")]
        assert keys == [k for k, _ in O.head_keys(cfg, 10)]
    m.freeze()
    assert not any(p.requires_grad for p in m.features.parameters())
    m.unfreeze()
    assert all(p.requires_grad for p in m.features.parameters())
    with pytest.raises(ValueError):
        FineTuneModelPool(base, "mnasnet", 10, "nope")
