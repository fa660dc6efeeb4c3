"""CPU oracle for the MNASNet hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The product path (``mnasnet_pytorch_amd``) never imports it and has no CPU fallback.

What it is: a plain eager-PyTorch fp32 *functional* restatement (``F.conv2d`` / ``F.batch_norm`` /
``relu``; explicit weight sharing; autograd for the backward) of the reference's hot path:

    ConvBlock      /root/reference/src/models/mnasnet.py:37-62
    SepConv        /root/reference/src/models/mnasnet.py:64-103
    MBConv_block   /root/reference/src/models/mnasnet.py:105-137
    MBConv         /root/reference/src/models/mnasnet.py:139-173
    Mnasnet        /root/reference/src/models/mnasnet.py:175-213
    FineTuneModelPool.forward   /root/reference/src/models/classifiers.py:19-111
    train() step body           /root/reference/src/train.py:423-440

The arithmetic itself lives in a third-party dependency of the reference (PyTorch ATen: conv2d,
native_batch_norm, relu; the reference pins only "PyTorch 0.4" in README.md:89-90, this image has torch
2.10.0).  The reference has no tests and no golden vectors of its own, so parity is pinned by fixtures
generated HERE by importing the reference itself (``tests/golden/make_golden.py``; the fixtures under
``tests/golden/*.npz`` hold only inputs/outputs, never reference source) -- ``tests/test_oracle_golden.py``
checks this restatement against every one of them.

Weights/inputs for every parity case come from the closed-form ``det_*`` fills below (no torch RNG), so
the GPU box can regenerate them without the reference.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5       # nn.BatchNorm2d default; mnasnet.py:55 passes nothing else
BN_MOMENTUM = 0.1   # the ConvBlock(momentum=) ctor argument is never forwarded (mnasnet.py:46,55)

# (in, out, channel_factor t, layers, kernel, reduce)  -- mnasnet.py:181-192
STAGES = [
    (16, 24, 3, 3, 3, True),
    (24, 40, 3, 3, 5, True),
    (40, 80, 6, 3, 5, True),
    (80, 96, 6, 2, 3, False),
    (96, 192, 6, 4, 5, True),
    (192, 320, 6, 1, 3, False),
]


# --------------------------------------------------------------------------------------------------
# deterministic closed-form fills (splitmix64 of the element index; no RNG state)
# --------------------------------------------------------------------------------------------------
def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def det_uniform(shape, seed: int) -> torch.Tensor:
    """float32 tensor, uniform in [-1, 1), element i = f(seed, i) only."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + (np.uint64(seed & 0xFFFFFFFF) << np.uint64(32))
        bits = _splitmix64(idx) >> np.uint64(40)            # 24 random bits
    u = bits.astype(np.float64) / float(1 << 24) * 2.0 - 1.0
    return torch.from_numpy(u.astype(np.float32).reshape(shape))


def key_seed(key: str, seed: int = 0) -> int:
    return (zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0xFFFFFFFF


def det_param(key: str, shape, seed: int = 0) -> torch.Tensor:
    """Deterministic value for one state_dict entry, chosen by the key suffix.

    conv.weight : uniform with the variance of kaiming_normal_(fan_out) (mnasnet.py:200)
    conv.bias   : 0.05*u   (non-zero on purpose: exercises the bias-before-BN path)
    bn.weight   : 1 + 0.2*u ; bn.bias : 0.1*u ; running_mean : 0.1*u ; running_var : 1 + 0.3*|u|
    """
    u = det_uniform(shape, key_seed(key, seed))
    if key.endswith("conv.weight"):
        fan_out = shape[0] * shape[2] * shape[3]
        return u * float(np.sqrt(2.0 / fan_out) * np.sqrt(3.0))
    if key.endswith("conv.bias"):
        return u * 0.05
    if key.endswith("bn.weight"):
        return 1.0 + 0.2 * u
    if key.endswith("bn.bias"):
        return 0.1 * u
    if key.endswith("running_mean"):
        return 0.1 * u
    if key.endswith("running_var"):
        return 1.0 + 0.3 * u.abs()
    if key.endswith("num_batches_tracked"):
        return torch.zeros((), dtype=torch.int64)
    if key.endswith(".weight"):   # classifier Linear
        return u * float(1.0 / np.sqrt(shape[1]))
    if key.endswith(".bias"):
        return u * 0.05
    raise KeyError(key)


# --------------------------------------------------------------------------------------------------
# topology: the ordered list of ConvBlock applications of one forward, with aliases
# --------------------------------------------------------------------------------------------------
class ConvSpec:
    __slots__ = ("prefix", "aliases", "cin", "cout", "k", "stride", "pad", "groups")

    def __init__(self, prefix, cin, cout, k, stride, pad, groups):
        self.prefix, self.cin, self.cout, self.k = prefix, cin, cout, k
        self.stride, self.pad, self.groups = stride, pad, groups
        self.aliases = [prefix]

    @property
    def kind(self):
        if self.groups == 1 and self.k == 1:
            return "pw"
        if self.groups == self.cin and self.groups > 1:
            return "dw"
        return "dense"

    def weight_shape(self):
        return (self.cout, self.cin // self.groups, self.k, self.k)


class SESpec:
    """Squeeze-and-excitation stage of the SE variant (BASELINE config 4).  BUILD-DEFINED: the reference has no SE block, so this
    restates mnasnet_pytorch_amd.mnasnet.SqueezeExcite, not a reference line ("parity unpinned by the reference")."""
    __slots__ = ("prefix", "aliases", "channels", "reduced")
    kind = "se"

    def __init__(self, prefix, channels, reduced):
        self.prefix, self.channels, self.reduced = prefix, channels, reduced
        self.aliases = [prefix]


def se_reduced(c: int, se_ratio: float) -> int:
    return max(8, int(round(c * se_ratio)))


def _block_specs(prefix: str, c: int, t: int, k: int, se_ratio: float = 0.0) -> list:
    """MBConv_block: 1x1 C->tC, kxk depthwise tC, 1x1 tC->C (mnasnet.py:116-129).  With se_ratio > 0 a 4th entry (SESpec) follows;
    it is applied BETWEEN the depthwise and the projection ConvBlocks."""
    specs = [
        ConvSpec(prefix + ".sequence.0", c, c * t, 1, 1, 0, 1),
        ConvSpec(prefix + ".sequence.1", c * t, c * t, k, 1, k // 2, c * t),
        ConvSpec(prefix + ".sequence.2", c * t, c, 1, 1, 0, 1),
    ]
    if se_ratio and se_ratio > 0:
        specs.append(SESpec(prefix + ".se", c * t, se_reduced(c, se_ratio)))
    return specs


def se_apply(a, st, se: SESpec):
    """a * sigmoid(fc2(relu(fc1(mean_hw a))))  -- mnasnet_pytorch_amd.mnasnet.SqueezeExcite (build-defined)."""
    p = se.prefix
    z = a.mean((2, 3))
    h = F.relu(F.linear(z, st[p + ".fc1.weight"], st[p + ".fc1.bias"]))
    s = torch.sigmoid(F.linear(h, st[p + ".fc2.weight"], st[p + ".fc2.bias"]))
    return a * s[:, :, None, None]


def build_program(ccf: bool, kernel: Optional[int] = None, se_ratio: float = 0.0):
    """Returns (program, unique_specs).

    program is a list of steps, each one of
        ("conv", spec)                      -- a ConvBlock application
        ("block", [spec_e, spec_d, spec_p]) -- an MBConv_block application (x + seq(x), mnasnet.py:133)
    Shared blocks appear ``layers`` times with the SAME spec objects (mnasnet.py:162-164 list-multiply).
    """
    prog = []
    uniq: List[ConvSpec] = []

    def add(spec):
        uniq.append(spec)
        return spec

    prog.append(("conv", add(ConvSpec("features.0", 3, 32, 3, 2, 1, 1))))                  # :179
    prog.append(("conv", add(ConvSpec("features.1.sequence.0", 32, 32, 3, 1, 1, 32))))     # :86-91 (repeat=0)
    prog.append(("conv", add(ConvSpec("features.1.sequence.1", 32, 16, 1, 1, 0, 1))))      # :92-95
    for si, (cin, cout, t, layers, k, reduce) in enumerate(STAGES):
        f = "features.%d" % (si + 2)
        stride = 2 if reduce else 1
        if kernel is not None:
            k = kernel            # config 4: every MBConv stage with k x k depthwise convs
        if ccf:   # [reduce conv] + [block]*layers, block at out width (mnasnet.py:150-153,157-166)
            prog.append(("conv", add(ConvSpec(f + ".sequence.0", cin, cout, 3, stride, 1, 1))))
            blk = [add(s) for s in _block_specs(f + ".sequence.1", cout, t, k, se_ratio)]
            for li in range(layers):
                if li > 0:
                    for j, s in enumerate(blk):
                        s.aliases.append("%s.sequence.%d.se" % (f, 1 + li) if s.kind == "se" else
                                         "%s.sequence.%d.sequence.%d" % (f, 1 + li, j))
                prog.append(("block", blk))
        else:     # reversed: [block]*layers at in width, then the 3x3 conv (mnasnet.py:167-168)
            blk = [add(s) for s in _block_specs(f + ".sequence.0", cin, t, k, se_ratio)]
            for li in range(layers):
                if li > 0:
                    for j, s in enumerate(blk):
                        s.aliases.append("%s.sequence.%d.se" % (f, li) if s.kind == "se" else
                                         "%s.sequence.%d.sequence.%d" % (f, li, j))
                prog.append(("block", blk))
            prog.append(("conv", add(ConvSpec("%s.sequence.%d" % (f, layers), cin, cout, 3, stride, 1, 1))))
    return prog, uniq


_SUFFIXES = ("conv.weight", "conv.bias", "bn.weight", "bn.bias", "bn.running_mean", "bn.running_var",
             "bn.num_batches_tracked")
_SE_SUFFIXES = ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")


def state_keys(ccf: bool, kernel: Optional[int] = None, se_ratio: float = 0.0) -> List[str]:
    """All state_dict keys in the reference's order (399 of them); with se_ratio > 0 the SE variant's extra
    ``...se.fc{1,2}.{weight,bias}`` keys follow (their position is not part of any contract)."""
    _, uniq = build_program(ccf, kernel, se_ratio)
    entries = []
    se_entries = []
    for s in uniq:
        for a in s.aliases:
            if s.kind == "se":
                se_entries.append(a)
            else:
                entries.append((a, s))
    # reference order = module registration order = lexicographic by numeric path components
    def sort_key(e):
        return [int(p) if p.isdigit() else -1 for p in e[0].split(".")]
    entries.sort(key=sort_key)
    return ["%s.%s" % (a, suf) for a, _ in entries for suf in _SUFFIXES] + \
        ["%s.%s" % (a, suf) for a in se_entries for suf in _SE_SUFFIXES]


def init_state(ccf: bool, seed: int = 0, dtype=torch.float32, proj_gamma: float = 1.0, kernel: Optional[int] = None,
               se_ratio: float = 0.0) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic state dict with every alias key present and aliases sharing storage.

    proj_gamma scales the BatchNorm weight of every MBConv_block's projection conv (``...sequence.2.bn.weight``).
    With 1.0 the untrained network amplifies a relative input perturbation ~110x by its output (measured:
    every block's un-damped ReLU(BN(.)) residual branch adds gain), which makes ANY reduced-precision
    implementation look 40 % off at the output; 0.1 gives a well-conditioned network (gain ~5) on which
    whole-network parity against the fp32 reference is meaningful."""
    _, uniq = build_program(ccf, kernel, se_ratio)
    by_alias = {}
    for s in uniq:
        if s.kind == "se":
            vals = {"fc1.weight": det_param(s.prefix + ".fc1.weight", (s.reduced, s.channels), seed),
                    "fc1.bias": det_param(s.prefix + ".fc1.bias", (s.reduced,), seed),
                    "fc2.weight": det_param(s.prefix + ".fc2.weight", (s.channels, s.reduced), seed),
                    "fc2.bias": det_param(s.prefix + ".fc2.bias", (s.channels,), seed)}
            for a in s.aliases:
                by_alias[a] = vals
            continue
        vals = {
            "conv.weight": det_param(s.prefix + ".conv.weight", s.weight_shape(), seed),
            "conv.bias": det_param(s.prefix + ".conv.bias", (s.cout,), seed),
            "bn.weight": det_param(s.prefix + ".bn.weight", (s.cout,), seed) *
            (proj_gamma if (s.kind == "pw" and s.prefix.endswith(".sequence.2") and s.prefix.count("sequence") == 2) else 1.0),
            "bn.bias": det_param(s.prefix + ".bn.bias", (s.cout,), seed),
            "bn.running_mean": det_param(s.prefix + ".bn.running_mean", (s.cout,), seed),
            "bn.running_var": det_param(s.prefix + ".bn.running_var", (s.cout,), seed),
            "bn.num_batches_tracked": torch.zeros((), dtype=torch.int64),
        }
        for a in s.aliases:
            by_alias[a] = vals
    out = OrderedDict()
    for k in state_keys(ccf, kernel, se_ratio):
        a, suf = k.rsplit(".", 2)[0], ".".join(k.rsplit(".", 2)[1:])
        out[k] = by_alias[a][suf]
    return out


# --------------------------------------------------------------------------------------------------
# functional forward
# --------------------------------------------------------------------------------------------------
def convblock(x, st, spec: ConvSpec, train: bool):
    """relu(bn(conv(x)+bias))  -- mnasnet.py:58-62. Updates running stats in ``st`` when train."""
    p = spec.prefix
    y = F.conv2d(x, st[p + ".conv.weight"], st[p + ".conv.bias"], stride=spec.stride, padding=spec.pad,
                 groups=spec.groups)
    y = F.batch_norm(y, st[p + ".bn.running_mean"], st[p + ".bn.running_var"], st[p + ".bn.weight"],
                     st[p + ".bn.bias"], training=train, momentum=BN_MOMENTUM, eps=BN_EPS)
    if train:
        st[p + ".bn.num_batches_tracked"] += 1
    return F.relu(y)


def mbconv_block(x, st, specs, train: bool):
    """input + sequence(input)  -- mnasnet.py:131-137 (ReLU after the projection too)."""
    h = x
    for j, s in enumerate(specs[:3]):
        if j == 2 and len(specs) == 4:
            h = se_apply(h, st, specs[3])         # SE variant (build-defined): between depthwise and projection
        h = convblock(h, st, s, train)
    return x + h


def features_forward(x, st, ccf: bool, train: bool, taps: Optional[dict] = None, kernel: Optional[int] = None,
                     se_ratio: float = 0.0):
    """Mnasnet.features(x)  -- mnasnet.py:211-213.  ``taps`` (optional dict) receives the output of
    every program step, keyed by step index, for layer-by-layer debugging."""
    prog, _ = build_program(ccf, kernel, se_ratio)
    h = x
    for i, (op, arg) in enumerate(prog):
        h = convblock(h, st, arg, train) if op == "conv" else mbconv_block(h, st, arg, train)
        if taps is not None:
            taps[i] = h
    return h


# --------------------------------------------------------------------------------------------------
# classifier head (classifiers.py:56-89) + the train step (train.py:423-440)
# --------------------------------------------------------------------------------------------------
HEAD_CONFIGS = {
    # name: list of (kind, args)   -- classifiers.py:56-89
    "256": [("drop", 0.5), ("lin", 320, 256), ("relu",), ("drop", 0.5), ("lin", 256, None)],
    "512_256": [("drop", 0.5), ("lin", 320, 512), ("relu",), ("drop", 0.5), ("lin", 512, 256), ("relu",),
                ("drop", 0.5), ("lin", 256, None)],
    "320": [("drop", 0.2), ("lin", 320, None)],
    "512": [("drop", 0.5), ("lin", 320, 512), ("relu",), ("drop", 0.5), ("lin", 512, None)],
}


def head_keys(config: str, num_classes: int) -> List[Tuple[str, tuple]]:
    out = []
    for i, layer in enumerate(HEAD_CONFIGS[config]):
        if layer[0] == "lin":
            o = layer[2] if layer[2] is not None else num_classes
            out.append(("classifier.%d.weight" % i, (o, layer[1])))
            out.append(("classifier.%d.bias" % i, (o,)))
    return out


def init_head_state(config: str, num_classes: int, seed: int = 0):
    return OrderedDict((k, det_param(k, shp, seed)) for k, shp in head_keys(config, num_classes))


def head_forward(f, hst, config: str, train: bool, dropout: bool = True):
    """classifier(AdaptiveAvgPool2d(1)(f).view(N,-1))  -- classifiers.py:107-111."""
    h = F.adaptive_avg_pool2d(f, 1).view(f.size(0), -1)
    for i, layer in enumerate(HEAD_CONFIGS[config]):
        if layer[0] == "drop":
            h = F.dropout(h, layer[1], training=train and dropout)
        elif layer[0] == "relu":
            h = F.relu(h)
        else:
            h = F.linear(h, hst["classifier.%d.weight" % i], hst["classifier.%d.bias" % i])
    return h


def head_dropout_keep(seed: int, n: int, p: float) -> np.ndarray:
    """keep mask (bool[n]) of an nn.Dropout(p) over n elements as the HIP head draws it (csrc/mnas_head.hip head_keep):
    element i is kept iff the high 32 bits of splitmix64(seed + golden*(i+1)) are >= p*2^32, so P(keep) = 1-p exactly as
    F.dropout's Bernoulli(1-p) (classifiers.py:57-88); kept elements are scaled by 1/(1-p)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + np.uint64(0x9E3779B97F4A7C15) * (idx + np.uint64(1))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    t = min(int(float(np.float32(p)) * 4294967296.0), 4294967295)
    return (z >> np.uint64(32)) >= np.uint64(t)


def head_forward_masked(h, hst, config: str, keeps: Optional[list]):
    """head_forward on already pooled features h (N, 320) with EXPLICIT dropout keep masks: keeps[j] (bool tensor shaped
    like the j-th Dropout's input, or None = no dropout) -- F.dropout semantics x * keep / (1-p)."""
    j = 0
    for i, layer in enumerate(HEAD_CONFIGS[config]):
        if layer[0] == "drop":
            k = keeps[j] if keeps is not None else None
            j += 1
            if k is not None:
                h = h * k.to(h.dtype) * (1.0 / (1.0 - layer[1]))
        elif layer[0] == "relu":
            h = F.relu(h)
        else:
            h = F.linear(h, hst["classifier.%d.weight" % i], hst["classifier.%d.bias" % i])
    return h


class OracleNet(torch.nn.Module):
    """nn.Module wrapper so torch.optim can drive the functional oracle (FineTuneModelPool over
    Mnasnet(ccf) -- what train.py:194-207 builds)."""

    def __init__(self, ccf=False, head: Optional[str] = "512", num_classes=1000, seed=0, kernel=None, se_ratio=0.0):
        super().__init__()
        self.ccf, self.head, self.num_classes = ccf, head, num_classes
        self.kernel, self.se_ratio = kernel, se_ratio
        st = init_state(ccf, seed, kernel=kernel, se_ratio=se_ratio)
        self._keys = list(st.keys())
        self._names = {}
        seen = {}
        for k, v in st.items():
            if id(v) in seen:
                self._names[k] = seen[id(v)]
                continue
            name = k.replace(".", "_")
            seen[id(v)] = name
            self._names[k] = name
            if v.dtype.is_floating_point and ("running" not in k):
                self.register_parameter(name, torch.nn.Parameter(v.clone()))
            else:
                self.register_buffer(name, v.clone())
        self._hnames = {}
        if head is not None:
            for k, v in init_head_state(head, num_classes, seed).items():
                name = k.replace(".", "_")
                self._hnames[k] = name
                self.register_parameter(name, torch.nn.Parameter(v.clone()))

    def state(self):
        return {k: getattr(self, n) for k, n in self._names.items()}

    def head_state(self):
        return {k: getattr(self, n) for k, n in self._hnames.items()}

    def features(self, x, taps=None):
        return features_forward(x, self.state(), self.ccf, self.training, taps, self.kernel, self.se_ratio)

    def forward(self, x, dropout=True):
        f = self.features(x)
        if self.head is None:
            return f
        return head_forward(f, self.head_state(), self.head, self.training, dropout)


def train_step(model: torch.nn.Module, optimizer, x, target, criterion=None):
    """One iteration of train.py:427-440 (no .item() syncs)."""
    criterion = criterion or torch.nn.CrossEntropyLoss()
    out = model(x.float())
    loss = criterion(out, target)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss


# --------------------------------------------------------------------------------------------------
# bf16 storage emulation: what the HIP path keeps in HBM (used by GPU parity tests for tight bounds)
# --------------------------------------------------------------------------------------------------
def round_bf16(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)
