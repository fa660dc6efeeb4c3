"""Case definitions shared by the golden generator (runs the REFERENCE, this container only) and the
parity tests (run the oracle on CPU / the HIP path on the GPU).  Pure data + closed-form fills: nothing
here comes from the reference's sources."""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import mnasnet_oracle as O  # noqa: E402

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))

# name: (cin, cout, k, stride, pad, groups, N, H, W)   -- SURVEY 8(c)(1)
PRIMITIVES = {
    "pw_16_48": (16, 48, 1, 1, 0, 1, 2, 12, 12),
    "pw_72_24": (72, 24, 1, 1, 0, 1, 2, 12, 12),
    "pw_96_576": (96, 576, 1, 1, 0, 1, 2, 6, 7),
    "dw3_48": (48, 48, 3, 1, 1, 48, 2, 12, 12),
    "dw5_72": (72, 72, 5, 1, 2, 72, 2, 12, 12),
    "dw5_240": (240, 240, 5, 1, 2, 240, 2, 7, 9),
    "dense_3_32_s2": (3, 32, 3, 2, 1, 1, 2, 12, 12),
    "dense_16_24_s2": (16, 24, 3, 2, 1, 1, 2, 12, 12),
    "dense_80_96_s1": (80, 96, 3, 1, 1, 1, 2, 7, 9),
}
# name: (C, t, k, N, H, W)   -- SURVEY 8(c)(2)
BLOCKS = {
    "block_16_3_3": (16, 3, 3, 2, 14, 14),
    "block_24_3_5": (24, 3, 5, 2, 14, 14),
    "block_96_6_5": (96, 6, 5, 2, 14, 14),
}
# name: (cin, cout, t, layers, k, reduce, ccf, N, H, W)  -- SURVEY 8(c)(3)
STAGES = {
    "stage_16_24_ccfT": (16, 24, 3, 3, 3, True, True, 2, 16, 16),
    "stage_16_24_ccfF": (16, 24, 3, 3, 3, True, False, 2, 16, 16),
}
# name: (cin, cout, k, reduce, repeat, N, H, W)   -- SepConv (mnasnet.py:64-103) as its own parity row.
# repeat = r puts the SAME (depthwise, pointwise) pair r times in front of the final pair (list-multiply, mnasnet.py:76-85):
# repeat >= 2 is the case in which weights, BatchNorm parameters and running statistics are shared between applications.
SEPCONVS = {
    "sep_32_16": (32, 16, 3, False, 0, 2, 12, 12),           # the network's own SepConv (mnasnet.py:180)
    "sep_16_16_rep1": (16, 16, 3, False, 1, 2, 12, 12),
    "sep_24_32_rep2_k5": (24, 32, 5, False, 2, 2, 10, 12),
    # reduce=True: the depthwise convs run with STRIDE 2 (mnasnet.py:73-75) -- with repeat = 1 twice in a row (12x13 -> 6x7 -> 3x4)
    "sep_32_16_reduce": (32, 16, 3, True, 0, 2, 12, 12),
    "sep_16_24_rep1_reduce_k5": (16, 24, 5, True, 1, 3, 12, 13),
}
# name: (ccf, N, H, W, train, proj_gamma)   -- SURVEY 8(c)(4); proj_gamma: see oracle.init_state
NETS = {
    "net_ccfT_64_train": (True, 2, 64, 64, True, 1.0),
    "net_ccfF_64_train": (False, 2, 64, 64, True, 1.0),
    "net_ccfF_rect_train": (False, 2, 96, 64, True, 1.0),
    # well-conditioned whole-network cases (gain ~5 instead of ~110) with >= 72 samples per channel in the
    # last stage: the ones on which bf16-storage results are comparable with the fp32 reference
    "net_ccfF_b8_96_wc_train": (False, 8, 96, 96, True, 0.1),
    "net_ccfT_b8_96_wc_train": (True, 8, 96, 96, True, 0.1),
    "net_ccfF_rect_wc_train": (False, 6, 96, 128, True, 0.1),
    "net_ccfT_224_eval": (True, 1, 224, 224, False, 1.0),    # BASELINE.json configs[0]
    "net_ccfF_224_eval": (False, 1, 224, 224, False, 1.0),
}


def sepconv_specs(name, prefix="sep"):
    """(program, unique specs) of one stand-alone SepConv: module index j of `sequence` -> ConvSpec with prefix
    '<prefix>.sequence.<j>'; the repeated pair is ONE spec pair applied `repeat` times (aliases = its other module indices)."""
    cin, cout, k, reduce, repeat, N, H, W = SEPCONVS[name]
    stride = 2 if reduce else 1
    prog, uniq = [], []
    if repeat > 0:
        dwa = O.ConvSpec("%s.sequence.0" % prefix, cin, cin, k, stride, k // 2, cin)
        pwa = O.ConvSpec("%s.sequence.1" % prefix, cin, cin, 1, 1, 0, 1)
        uniq += [dwa, pwa]
        for r in range(repeat):
            if r > 0:
                dwa.aliases.append("%s.sequence.%d" % (prefix, 2 * r))
                pwa.aliases.append("%s.sequence.%d" % (prefix, 2 * r + 1))
            prog += [("conv", dwa), ("conv", pwa)]
    dwb = O.ConvSpec("%s.sequence.%d" % (prefix, 2 * repeat), cin, cin, k, stride, k // 2, cin)
    pwb = O.ConvSpec("%s.sequence.%d" % (prefix, 2 * repeat + 1), cin, cout, 1, 1, 0, 1)
    uniq += [dwb, pwb]
    prog += [("conv", dwb), ("conv", pwb)]
    return prog, uniq


def sepconv_state(name, uniq, prefix="sep"):
    """Closed-form state of a stand-alone SepConv keyed like the generator's load_det: '<name>.sequence.<j>.<suffix>' of the
    FIRST alias."""
    import torch
    st = {}
    for s_ in uniq:
        tail = s_.prefix[len(prefix) + 1:]
        for suf, shp in (("conv.weight", s_.weight_shape()), ("conv.bias", (s_.cout,)), ("bn.weight", (s_.cout,)),
                         ("bn.bias", (s_.cout,)), ("bn.running_mean", (s_.cout,)), ("bn.running_var", (s_.cout,))):
            st[s_.prefix + "." + suf] = O.det_param("%s.%s.%s" % (name, tail, suf), shp, STATE_SEED)
        st[s_.prefix + ".bn.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    return st


HEADS = ("256", "512_256", "320", "512")
STATE_SEED = 1
INPUT_SEED = 7
COT_SEED = 11


def det_input(shape, seed=INPUT_SEED):
    return O.det_uniform(shape, seed)


def cotangent(shape, seed=COT_SEED):
    """dL/d(out): L = sum(out * cotangent) gives a non-trivial, reproducible backward."""
    return O.det_uniform(shape, seed)


def summarize(t):
    """(sum, sum|.|, sum of squares) in float64 -- the 'checksum' stored for large tensors."""
    d = t.detach().double()
    return [float(d.sum()), float(d.abs().sum()), float((d * d).sum())]
