"""bf16-storage mirror of the oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Same arithmetic as oracle/mnasnet_oracle.py (i.e. as the reference: ConvBlock mnasnet.py:58-62, MBConv_block
:131-137, autograd's batch-norm / ReLU / conv backward formulas), restated with an EXPLICIT backward and with
a round-to-bf16 at exactly the points where the HIP path stores a tensor in HBM or stages it into LDS:
  * activations consumed by a 1x1/3x3 conv: relu(s*y+t) -> bf16 (depthwise convs apply it on the fly in fp32)
  * raw conv output y -> bf16 (statistics from fp32)
  * 1x1 / 3x3 weights -> bf16 (depthwise weights stay fp32)   * residual sum r -> bf16
  * incoming/outgoing activation gradients g -> bf16          * dy = c1*dz + c2*y + c3 -> bf16 (fp32 in depthwise)
Everything else is fp32 (fp64 for the per-channel reductions, like the finalize kernels).

Why it exists: against the fp32 oracle a bf16 pipeline legitimately differs by a few % in L2 (ReLU-mask
flips of elements whose pre-activation is within bf16 rounding of zero turn a 0.4 % perturbation into an
O(1) change of single gradient elements).  Against THIS mirror the HIP engine must agree to ~1e-3, which is
what pins the engine's wiring (buffers, ordering, coefficient formulas, weight sharing) tightly; the mirror
itself is pinned to the fp32 oracle/goldens within the loose bf16 tolerance by tests/test_bf16_mirror.py.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .mnasnet_oracle import BN_EPS, BN_MOMENTUM, ConvSpec, round_bf16


class MAct:
    """(possibly virtual) activation: value = relu(s*data+t) if s is not None else data. data holds bf16 values."""

    def __init__(self, data, s=None, t=None):
        self.data, self.s, self.t = data, s, t

    def f32(self):
        if self.s is None:
            return self.data
        return F.relu(self.data * self.s.view(1, -1, 1, 1) + self.t.view(1, -1, 1, 1))

    def staged(self):
        return self.data if self.s is None else round_bf16(self.f32())


def _kind(spec: ConvSpec):
    return spec.kind


def conv_fwd(spec: ConvSpec, a_in, st, train, image=None):
    p = spec.prefix
    if image is not None:
        a = round_bf16(image)
    else:
        # depthwise: act-on-read in fp32; the MFMA kernels stage the activation as bf16
        a = a_in.f32() if _kind(spec) == "dw" else a_in.staged()
    W = st[p + ".conv.weight"].detach()
    w = W if _kind(spec) == "dw" else round_bf16(W)
    y32 = F.conv2d(a, w, st[p + ".conv.bias"].detach(), stride=spec.stride, padding=spec.pad, groups=spec.groups)
    gamma, beta = st[p + ".bn.weight"].detach().double(), st[p + ".bn.bias"].detach().double()
    M = y32.numel() // y32.shape[1]
    if train:
        y64 = y32.double()
        mean = y64.mean((0, 2, 3))
        var = (y64 * y64).mean((0, 2, 3)) - mean * mean
        var = var.clamp_min(0)
        invstd = 1.0 / torch.sqrt(var + BN_EPS)
        rm, rv = st[p + ".bn.running_mean"], st[p + ".bn.running_var"]
        rm.copy_(((1 - BN_MOMENTUM) * rm.double() + BN_MOMENTUM * mean).float())
        rv.copy_(((1 - BN_MOMENTUM) * rv.double() + BN_MOMENTUM * var * M / max(M - 1, 1)).float())
        st[p + ".bn.num_batches_tracked"] += 1
    else:
        mean = st[p + ".bn.running_mean"].double()
        invstd = 1.0 / torch.sqrt(st[p + ".bn.running_var"].double() + BN_EPS)
    s = (gamma * invstd).float()
    t = (beta - mean * gamma * invstd).float()
    y = round_bf16(y32)
    out = MAct(y, s, t)
    saved = dict(spec=spec, a=a, w=w, w32=W, y=y, s=s, t=t, mean=mean.float(), invstd=invstd.float(), M=M,
                 in_shape=tuple(a.shape), image=image is not None)
    return out, saved


def conv_bwd(saved, g, grads, resid=None, need_gin=True, se=None):
    """g: bf16-valued grad wrt the activated output.  Accumulates parameter grads into ``grads`` (dict keyed by
    state_dict names).  Returns bf16-valued grad wrt the activated input (or None).
    se (the saved dict of se_fwd, excitation applied ON LOAD: Engine.se_on_load): this is the project conv of a squeeze-excite
    block -- its weight gradient is formed per image on the UNGATED bf16 activation and gated in fp32, and the same per-image
    sums give dL/du without a pass over (gs, a2) (csrc/mnas_se.hip k_se_proj_du); stored into se["du"]."""
    spec, a, w, y, s, t = saved["spec"], saved["a"], saved["w"], saved["y"], saved["s"], saved["t"]
    mean, invstd, M = saved["mean"], saved["invstd"], saved["M"]
    v = lambda c: c.view(1, -1, 1, 1)
    dz = g * ((y * v(s) + v(t)) > 0)
    xhat = y * v(invstd) + v(-mean * invstd)
    S1 = dz.double().sum((0, 2, 3))
    S2 = (dz * xhat).double().sum((0, 2, 3))
    sd, isd, md = s.double(), invstd.double(), mean.double()
    c1 = s
    c2 = (-sd * isd * S2 / M).float()
    c3 = (sd * (md * isd * S2 / M - S1 / M)).float()
    dy = v(c1) * dz + (v(c2) * y + v(c3))
    if _kind(spec) != "dw":
        dy = round_bf16(dy)                 # staged into LDS as bf16 for the MFMA kernels; the
                                            # per-layer depthwise kernels keep fp32
    p = spec.prefix

    def acc(name, val):
        grads[name] = grads.get(name, 0) + val

    acc(p + ".bn.weight", S2.float())
    acc(p + ".bn.bias", S1.float())
    acc(p + ".conv.bias", torch.zeros(spec.cout))
    if se is not None:
        sg = se["sg"]
        Pn = torch.einsum("nohw,nchw->noc", dy.double(), round_bf16(se["a"]).double())        # per-image dy^T a2 (fp32 accumulate)
        acc(p + ".conv.weight", (Pn * sg[:, None, :].double()).sum(0).float().view(tuple(w.shape)))
        W32 = saved["w32"].double().view(1, w.shape[0], w.shape[1])
        se["du"] = ((Pn * W32).sum(1) * (sg * (1 - sg)).double()).float()
    else:
        acc(p + ".conv.weight", torch.nn.grad.conv2d_weight(a, tuple(w.shape), dy, stride=spec.stride, padding=spec.pad,
                                                            groups=spec.groups))
    if not need_gin:
        return None
    gin = torch.nn.grad.conv2d_input(saved["in_shape"], w, dy, stride=spec.stride, padding=spec.pad, groups=spec.groups)
    if resid is not None:
        gin = gin + resid
    return gin if saved.get("image") else round_bf16(gin)          # dL/d image leaves the path as fp32 (csrc/mnas_stem.hip k_stem_dgrad)


def se_fwd(se, a2: "MAct", st):
    """Squeeze-excite of the SE variant with the HIP path's rounding points (csrc/mnas_se.hip): pooled mean, MLP and sigmoid in
    fp32 on the fp32 activation; the scaled activation is MATERIALISED as bf16 for the project conv."""
    p = se.prefix
    a = a2.f32()
    z = a.mean((2, 3))
    w1, b1 = st[p + ".fc1.weight"].detach(), st[p + ".fc1.bias"].detach()
    w2, b2 = st[p + ".fc2.weight"].detach(), st[p + ".fc2.bias"].detach()
    h = F.relu(F.linear(z, w1, b1))
    sg = torch.sigmoid(F.linear(h, w2, b2))
    out = MAct(round_bf16(a * sg[:, :, None, None]))
    return out, dict(se=se, a=a, z=z, h=h, sg=sg, w1=w1, w2=w2)


def se_bwd(saved, gs, grads):
    """gs: bf16-valued dL/d(a * s).  Accumulates the SE parameters' gradients; returns bf16-valued dL/da."""
    se, a, z, h, sg, w1, w2 = (saved[k] for k in ("se", "a", "z", "h", "sg", "w1", "w2"))
    HW = a.shape[2] * a.shape[3]
    du = saved["du"] if "du" in saved else (gs * a).sum((2, 3)) * sg * (1 - sg)      # "du": excitation on load (conv_bwd above)
    dh = (du @ w2) * (h > 0)
    dz = dh @ w1
    p = se.prefix

    def acc(name, val):
        grads[name] = grads.get(name, 0) + val
    acc(p + ".fc2.weight", du.t() @ h)
    acc(p + ".fc2.bias", du.sum(0))
    acc(p + ".fc1.weight", dh.t() @ z)
    acc(p + ".fc1.bias", dh.sum(0))
    ga = gs * sg[:, :, None, None] + dz[:, :, None, None] / HW
    return round_bf16(ga)                   # k_se_bwd_apply stores it as bf16


def run(program, st, x, train=True, cot=None, need_dx=False, se_on_load=None):
    """program: list of ("conv", spec) / ("block", [e,d,p]) (oracle.build_program or hand-made).
    se_on_load: None, or a predicate (N, H, W, E) -> bool naming the squeeze-excite blocks whose excitation the engine applies on
    load (Engine.se_on_load; the forward values are the same, the project conv's backward rounds differently: conv_bwd).
    Returns dict(y=fp32 output, grads={name: tensor}, dx=fp32 or None)."""
    first = program[0][1] if program[0][0] == "conv" else program[0][1][0]
    is_image = first.kind == "dense" and first.cin == 3
    cur = None if is_image else MAct(round_bf16(x))
    tape = []
    for op, arg in program:
        if op == "conv":
            cur, sv = conv_fwd(arg, cur, st, train, image=x if (cur is None) else None)
            tape.append(("conv", sv))
        else:
            a_in = cur
            h = cur
            svs = []
            N_, C_, H_, W_ = a_in.data.shape
            sse = None
            for j, spec in enumerate(arg[:3]):
                if j == 2 and len(arg) == 4:
                    h, sse = se_fwd(arg[3], h, st)
                    sse["on_load"] = bool(se_on_load and se_on_load(N_, H_, W_, h.data.shape[1]))
                h, sv = conv_fwd(spec, h, st, train)
                svs.append(sv)
            svs.append(sse)
            cur = MAct(round_bf16(a_in.f32() + h.f32()))
            tape.append(("block", svs))
    out = dict(y=cur.f32(), grads={}, dx=None)
    if cot is None:
        return out
    g = round_bf16(cot)
    grads = out["grads"]
    for n in range(len(tape) - 1, -1, -1):
        kind, sv = tape[n]
        first_step = n == 0
        if kind == "conv":
            need = (not first_step) or need_dx
            g = conv_bwd(sv, g, grads, None, need)
        else:
            G = g
            g2 = conv_bwd(sv[2], G, grads, se=sv[3] if (sv[3] is not None and sv[3]["on_load"]) else None)
            if sv[3] is not None:
                g2 = se_bwd(sv[3], g2, grads)
            g1 = conv_bwd(sv[1], g2, grads)
            need = (not first_step) or need_dx
            g = conv_bwd(sv[0], g1, grads, G if need else None, need)
    out["dx"] = g
    return out
