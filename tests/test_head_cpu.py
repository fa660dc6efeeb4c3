"""CPU: host logic of the native head (mnasnet_pytorch_amd/head.py) and the oracle's restatement of its dropout mask
(oracle.head_dropout_keep / head_forward_masked) -- classifiers.py:56-89."""
import numpy as np
import torch
import torch.nn as nn

import cases as C
from cases import O
from mnasnet_pytorch_amd.head import parse_sequential, _mix


def test_parse_reference_heads():
    """every classifier_config of classifiers.py:56-89 is a Dropout/Linear/ReLU chain the native head takes"""
    from mnasnet_pytorch_amd import FineTuneModelPool, load_model
    import contextlib, io
    for cfg in C.HEADS:
        with contextlib.redirect_stdout(io.StringIO()):
            m = FineTuneModelPool(load_model("mnasnet"), "mnasnet", 10, cfg)
        layers = parse_sequential(m.classifier)
        spec = O.HEAD_CONFIGS[cfg]
        lins = [l for l in spec if l[0] == "lin"]
        assert layers is not None and len(layers) == len(lins)
        drops = [l[1] for l in spec if l[0] == "drop"]
        assert [l.p for l in layers] == drops
        relus = [i + 1 < len(spec) and spec[i + 1][0] == "relu" for i, l in enumerate(spec) if l[0] == "lin"]
        assert [l.relu for l in layers] == relus


def test_unsupported_heads_stay_pytorch():
    assert parse_sequential(nn.Sequential(nn.Linear(4, 4), nn.BatchNorm1d(4))) is None
    assert parse_sequential(nn.Sequential(nn.Linear(4, 4), nn.Dropout())) is None
    assert parse_sequential(nn.Sequential(nn.ReLU(), nn.Linear(4, 4))) is None
    assert parse_sequential(nn.Sequential(nn.Linear(4, 5), nn.ReLU(), nn.Linear(4, 3))) is None
    assert parse_sequential(nn.Sequential(nn.Dropout(1.0), nn.Linear(4, 4))) is None
    assert parse_sequential(nn.Linear(4, 4)) is None
    ok = parse_sequential(nn.Sequential(nn.Dropout(0.3), nn.Linear(4, 5), nn.ReLU(), nn.Linear(5, 3)))
    assert [(l.p, l.relu) for l in ok] == [(0.3, True), (0.0, False)]


def test_dropout_hash_statistics_and_determinism():
    for p in (0.2, 0.5):
        k1 = O.head_dropout_keep(_mix(123, 1, 0), 1 << 18, p)
        k2 = O.head_dropout_keep(_mix(123, 1, 0), 1 << 18, p)
        k3 = O.head_dropout_keep(_mix(123, 2, 0), 1 << 18, p)
        assert np.array_equal(k1, k2) and not np.array_equal(k1, k3)
        assert abs(k1.mean() - (1 - p)) < 5e-3
        # neighbouring elements are not correlated (a counter-based hash, not an LCG on the index)
        assert abs(np.corrcoef(k1[:-1], k1[1:])[0, 1]) < 1e-2
    assert O.head_dropout_keep(7, 1000, 0.0).all()


def test_masked_head_equals_plain_head_without_dropout():
    for cfg in C.HEADS:
        hst = O.init_head_state(cfg, 10, C.STATE_SEED)
        f = O.det_uniform((5, 320, 3, 3), 3).abs()
        ref = O.head_forward(f, hst, cfg, train=False)
        h = torch.nn.functional.adaptive_avg_pool2d(f, 1).view(5, -1)
        assert torch.allclose(O.head_forward_masked(h, hst, cfg, None), ref)
        ones = [torch.ones(5, l[1] if False else n, dtype=torch.bool) for l, n in
                zip([x for x in O.HEAD_CONFIGS[cfg] if x[0] == "drop"],
                    [x[1] for x in O.HEAD_CONFIGS[cfg] if x[0] == "lin"])]
        # all-ones masks = F.dropout in training mode with nothing dropped, scaled by 1/(1-p)
        out = O.head_forward_masked(h, hst, cfg, ones)
        assert out.shape == ref.shape and torch.isfinite(out).all()


def test_model_with_cached_native_head_pickles_and_deepcopies():
    """FineTuneModelPool caches a NativeHead (which holds the ctypes library handle) after the first forward: the cache must
    stay out of pickles and deep copies (torch.save(model), EMA / best-model copies)."""
    import contextlib, copy, io, pickle
    from mnasnet_pytorch_amd import FineTuneModelPool, load_model
    with contextlib.redirect_stdout(io.StringIO()):
        m = FineTuneModelPool(load_model("mnasnet"), "mnasnet", 10, "512")
    head = m._native_head()                 # loads libmnas_hip.so (no GPU call)
    assert head is not None and m._head is head
    m2 = copy.deepcopy(m)
    assert m2._head is None and m2._head_key is None
    assert m2.classifier[1].weight.data_ptr() != m.classifier[1].weight.data_ptr()
    m3 = pickle.loads(pickle.dumps(m))
    assert m3._head is None and list(m3.state_dict().keys()) == list(m.state_dict().keys())
    buf = io.BytesIO()
    torch.save(m, buf)
    assert m._head is head                  # the live model keeps its cache
    assert m2._native_head() is not None    # and a copy rebuilds its own lazily
