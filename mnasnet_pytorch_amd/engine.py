"""Execution engine: compiles a subtree of the drop-in modules (mnasnet.py) into a static PROGRAM of HIP
kernel launches (include/mnas.h) per (batch, height, width, mode) and runs it with ONE host->library call
per forward / per backward segment (mnas_run_ops_multi: main stream + an OPTIONAL side stream for weight-gradient kernels,
Engine.use_side_stream).

Mirrors, for this path, what autograd + ATen do for the reference:
    forward   Mnasnet.features(x)                /root/reference/src/models/mnasnet.py:211-213
    backward  loss.backward() through it          /root/reference/src/train.py:439
but MI355X-first: NHWC bf16 activations resident in HBM (nothing is recomputed: 288 GB), BatchNorm+ReLU
fused into the consumers' loads, BatchNorm statistics fused into the producers' epilogues, all buffers
allocated once per shape, no Python per-layer dispatch on the step path.

Autograd contract: parameters are inputs of one autograd.Function, so loss.backward() reaches
us (tensor hooks registered on parameters -- register_hook / post-accumulate-grad hooks -- do NOT fire: the gradients
bypass AccumulateGrad; DDP-style hook-driven reducers must use train_step.Trainer's stage-done callback instead); weight/BN gradients are written by the kernels straight into a flat fp32 buffer whose slices ARE the
parameters' ``.grad`` (None -> attached, already ours -> accumulated, foreign tensor -> added into it), the
same observable behaviour as autograd's AccumulateGrad, without ~110 tiny copy kernels.  Shared blocks
accumulate their ``layers`` contributions; ``conv.bias.grad`` is exactly zero (train-mode BatchNorm cancels
the bias; the reference gets ~1e-5 rounding noise there).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib as L

# igemm layers with fewer output pixels than this get one tile per workgroup (the 7x7 stage: 98 persistent workgroups of
# two tiles leave most of the 256 CUs idle; measured -23..-35 % per launch there, +17..+37 % on the 14x14 stage)
_SMALL_M = 20000
_STATS_PARTS = 2048          # persistent pixel-workgroups for conv kernels / rows of the stats scratch
_STEM_WGRAD_PARTS_MAX = 768  # upper bound of mnas_stem_parts(1, ...) (csrc/mnas_stem.hip): sizes the partial-slab scratch


def _cdiv(a, b):
    return (a + b - 1) // b


class _ConvInfo:
    """One unique ConvBlock (shared blocks appear once)."""

    def __init__(self, mod, stage):
        conv, bn = mod.conv, mod.bn
        self.mod, self.stage = mod, stage
        self.cin, self.cout = conv.in_channels, conv.out_channels
        self.k = conv.kernel_size[0]
        self.stride, self.pad, self.groups = conv.stride[0], conv.padding[0], conv.groups
        if conv.kernel_size[0] != conv.kernel_size[1] or conv.stride[0] != conv.stride[1]:
            raise NotImplementedError("square kernels / strides only")
        if bn.momentum is None or not bn.track_running_stats or not bn.affine:
            raise NotImplementedError("BatchNorm2d(affine, track_running_stats, momentum=float) only")
        if self.groups == 1 and self.k == 1:
            if self.stride != 1 or self.pad != 0:
                raise NotImplementedError("1x1 convs are stride 1 / pad 0 in this network")
            self.kind = "pw"
        elif self.groups == self.cin == self.cout and self.groups > 1:
            if self.k not in (3, 5) or self.stride not in (1, 2) or self.pad != self.k // 2:
                raise NotImplementedError("depthwise: k in {3,5}, stride 1 or 2, pad k//2 (mnasnet.py:73-81,122-125)")
            self.kind = "dw"
        elif self.groups == 1 and self.k == 3 and self.pad == 1 and self.cin == 3 and self.stride == 2:
            self.kind = "stem"
        elif self.groups == 1 and self.k == 3 and self.pad == 1:
            self.kind = "dense"
        else:
            raise NotImplementedError("unsupported ConvBlock geometry")
        if self.kind != "stem" and (self.cin % 8 or self.cout % 8):
            raise NotImplementedError("channel counts must be multiples of 8")
        self.params = [conv.weight, conv.bias, bn.weight, bn.bias]
        self.gslice = {}            # name -> (offset, numel) in the flat grad buffer

    def out_hw(self, H, W):
        return ((H + 2 * self.pad - self.k) // self.stride + 1, (W + 2 * self.pad - self.k) // self.stride + 1)


class _SEInfo:
    """One unique SqueezeExcite module (build-defined SE variant): parameter bookkeeping like _ConvInfo."""

    def __init__(self, mod, stage):
        self.mod, self.stage = mod, stage
        self.channels, self.reduced = mod.fc1.in_features, mod.fc1.out_features
        self.params = [mod.fc1.weight, mod.fc1.bias, mod.fc2.weight, mod.fc2.bias]
        self.gslice = {}


def se_segments_per_image(N, HW, tile, slices, max_slabs, resident=512):
    """Workgroups per image (a divisor of HW, so that no segment straddles two images) for the segment-mode backward of a
    squeeze-excite project conv: the divisor with the least (ragged-last-tile waste) x (idle share of the last round of `resident`
    workgroups), ties to the smaller one; 0 if no divisor keeps the N*d weight-gradient slabs within `max_slabs` / 4096."""
    best = None
    for d in range(1, 65):
        if HW % d or N * d > 4096 or N * d > max_slabs:
            continue
        seg = HW // d
        waste = _cdiv(seg, tile) * tile / seg            # pixel slots per pixel (ragged last tile of a segment)
        rounds = N * d * slices / float(resident)        # two resident workgroups per CU
        cost = waste * _cdiv(N * d * slices, resident) / rounds
        if best is None or cost < best[0] - 1e-9:
            best = (cost, d)
    return best[1] if best is not None else 0


def _trace(root, se_map=None):
    """Flatten a module subtree into steps: ("conv", ConvBlock, stage) / ("block", [e,d,p], stage).  se_map (optional dict)
    receives id(expand ConvBlock) -> SqueezeExcite module for blocks that carry one."""
    steps = []

    def rec(m, stage):
        name = type(m).__name__
        if name == "ConvBlock":
            steps.append(("conv", m, stage))
        elif name == "MBConv_block":
            steps.append(("block", list(m.sequence), stage))
            if se_map is not None and getattr(m, "se", None) is not None:
                se_map[id(m.sequence[0])] = m.se
        elif name in ("SepConv", "MBConv"):
            for c in m.sequence:
                rec(c, stage)
        elif name == "Mnasnet":
            rec(m.features, stage)
        elif isinstance(m, nn.Sequential):
            for i, c in enumerate(m):
                rec(c, i if stage is None else stage)
        else:
            raise TypeError("cannot compile %s for the HIP engine" % name)

    rec(root, None)
    return [(op, m, 0 if st is None else st) for op, m, st in steps]


class _Act:
    """A (possibly virtual) activation: value = relu(scale*data+shift) if bn is not None else data."""
    __slots__ = ("data", "bn", "H", "W", "C", "gate")

    def __init__(self, data, bn, H, W, C_, gate=None):
        self.data, self.bn, self.H, self.W, self.C = data, bn, H, W, C_
        self.gate = gate         # squeeze-excite applied on load: fp32 [N][C] multiplier on top of the virtual activation

    def act_ptrs(self):
        if self.bn is None:
            return [self.data.data_ptr(), None, None]
        return [self.data.data_ptr(), self.bn.data_ptr(), self.bn.data_ptr() + 4 * self.C]


class _OpList:
    def __init__(self, eng=None, tag=""):
        self.items = []
        self.eng, self.tag = eng, tag

    def add(self, opcode, ints=(), dbls=(), ptrs=(), stream=0):
        """stream: 0 = main (torch's current stream), 1 = the engine's side stream (weight-gradient work)"""
        prof = (self.eng is not None and self.eng.profile_opcodes and opcode in self.eng.profile_opcodes
                and (self.eng.profile_filter is None or self.eng.profile_filter(opcode, tuple(ints))))
        if prof:
            ev0, ev1 = self.eng.new_event(), self.eng.new_event()
            gate = C.addressof(self.eng.profile_gate)          # host int: 0 = these two records are skipped (Engine.profile_gate)
            self.items.append((L.OP_EVENT_RECORD, [], [], [ev0, gate], stream))
        self.items.append((opcode, list(ints), list(dbls), list(ptrs), stream))
        idx = len(self.items) - 1
        if prof:
            self.items.append((L.OP_EVENT_RECORD, [], [], [ev1, gate], stream))
            self.eng.profile_events.append(((self.tag, opcode, tuple(ints)), ev0, ev1))
        return idx

    def fork(self):
        """side stream may start from here: it waits for everything enqueued on main so far"""
        ev = self.eng.new_event()
        self.add(L.OP_EVENT_RECORD, [], [], [ev], 0)
        self.add(L.OP_EVENT_WAIT, [], [], [ev], 1)

    def join(self):
        """main waits for everything enqueued on the side stream so far"""
        ev = self.eng.new_event()
        self.add(L.OP_EVENT_RECORD, [], [], [ev], 1)
        self.add(L.OP_EVENT_WAIT, [], [], [ev], 0)

    def build(self):
        arr = (L.MnasOp * max(1, len(self.items)))()
        for n, (opc, ints, dbls, ptrs, stream) in enumerate(self.items):
            o = arr[n]
            o.opcode = opc
            o.i[14] = stream
            for j, v in enumerate(ints):
                o.i[j] = int(v)
            for j, v in enumerate(dbls):
                o.d[j] = float(v)
            for j, v in enumerate(ptrs):
                o.p[j] = v if v else None
        return arr, len(self.items)


class Program:
    """All buffers + launch lists for one (N, H, W, training, need_dx) configuration."""

    def __init__(self, eng: "Engine", N, H, W, training, need_dx, pooled=False, in_u8=False):
        self.eng, self.N, self.H, self.W, self.training, self.need_dx = eng, N, H, W, training, need_dx
        self.pooled = pooled
        self.in_u8 = bool(in_u8)
        # fused input pipeline of the stem (Engine.set_input_normalization): per-plane affine, uint8 images
        aff = eng.input_affine(self.in_u8)
        self._aff_ptr = aff.data_ptr() if aff is not None else None
        if self.in_u8 and aff is None:
            raise RuntimeError("uint8 images need Engine.set_input_normalization(mean, std)")
        self.busy = False
        dev = eng.device
        self.keep = []                      # tensors owned by this program
        lib = eng.lib



        fwd = self._fwd = _OpList(eng, "fwd")
        # dy plane (Ho, Wo) of every stride-2 dense conv in THIS program: the transposed-conv input gradient is picked per plane
        # (rectangular clusters, 192-px inputs ... get the form whenever the library has it for their plane)
        self._tconv_ok = {}
        Ht, Wt = H, W
        for op, m, stage in eng.steps:
            for cb in ([m] if op == "conv" else m):
                ci = eng.info[id(cb)]
                Ho_, Wo_ = ci.out_hw(Ht, Wt)
                if eng.use_tconv and eng.materialize_dy and getattr(ci, "w_tconv", None) is not None and Ht == 2 * Ho_ and Wt == 2 * Wo_:
                    self._tconv_ok[id(ci)] = bool(lib.mnas_tconv_supported(Ho_, Wo_, ci.cout, ci.cin)) and \
                        lib.mnas_tconv_parts(N, Ho_, Wo_, ci.cout, ci.cin) > 0
                Ht, Wt = Ho_, Wo_
        # ---- weight packing (once per forward; weights change every optimizer step): one batched launch
        descs = []
        for ci in eng.convs:
            w = ci.mod.conv.weight
            if ci.kind in ("pw", "dense"):
                descs.append((w.data_ptr(), ci.w_fwd.data_ptr(), L.PACK_FWD, ci.cout, ci.cin, ci.k * ci.k))
                if training:
                    descs.append((w.data_ptr(), ci.w_dgrad.data_ptr(), L.PACK_DGRAD, ci.cout, ci.cin, ci.k * ci.k))
                    if self._tconv_ok.get(id(ci)):
                        descs.append((w.data_ptr(), ci.w_tconv.data_ptr(), L.PACK_TCONV, ci.cout, ci.cin, 9))
            elif ci.kind == "dw":
                descs.append((w.data_ptr(), ci.w_fwd.data_ptr(), L.PACK_DW, ci.cout, 1, ci.k * ci.k))
            else:  # stem: [Co][27] viewed as a 1x1 conv over 27 "channels"
                descs.append((w.data_ptr(), ci.w_fwd.data_ptr(), L.PACK_FWD, ci.cout, 27, 1))
        host = (L.MnasPackDesc * len(descs))()
        for n_, (wp, dp, kind, co, cin_, taps) in enumerate(descs):
            host[n_].w, host[n_].dst, host[n_].kind, host[n_].Co, host[n_].Ci, host[n_].taps = wp, dp, kind, co, cin_, taps
        raw = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(dev)
        self.keep.append(raw)
        fwd.add(L.OP_PACK_BATCH, [len(descs)], [], [raw.data_ptr()])

        steps = eng.steps
        first_kind = eng.info[id(steps[0][1] if steps[0][0] == "conv" else steps[0][1][0])].kind
        self.x_is_image = first_kind == "stem"
        self.patch_x = []       # (op index, pointer slot) receiving the input pointer
        records = self._records = []   # forward applications, for the backward builder
        Hc, Wc = H, W
        if self.x_is_image:
            cur = None          # the stem reads the fp32 NCHW input directly
        else:
            Cin = eng.info[id(steps[0][1] if steps[0][0] == "conv" else steps[0][1][0])].cin
            xb = self._new((N, H, W, Cin))
            j = fwd.add(L.OP_NCHW_TO_NHWC, [N, Cin, H * W], [], [None, xb.data_ptr()])
            self.patch_x.append((j, 0))
            cur = _Act(xb, None, H, W, Cin)
        self.in_channels = 3 if self.x_is_image else cur.C





        self._se_records = {}    # record index of a project conv -> its block's squeeze-excite tensors
        step_records = []
        for op, m, stage in steps:
            start = len(records)
            if op == "conv":
                ci = eng.info[id(m)]
                cur = self._conv_fwd(ci, cur, Hc, Wc)
                Hc, Wc = cur.H, cur.W
                step_records.append(("conv", stage, start, None, None))
            else:
                a_in = cur
                cis = [eng.info[id(cb)] for cb in m]
                se = eng.se_info.get(id(m[0]))
                h = cur
                for j_, ci_ in enumerate(cis):
                    if j_ == 2 and se is not None:
                        h = self._se_fwd(se, h, Hc, Wc, ci_)
                    h = self._conv_fwd(ci_, h, Hc, Wc)
                r = self._new((N, Hc, Wc, a_in.C))
                fwd.add(L.OP_ADD_ACT, [a_in.C, Hc * Wc], [float(N * Hc * Wc)],
                        a_in.act_ptrs() + h.act_ptrs() + [r.data_ptr(), None])
                cur = _Act(r, None, Hc, Wc, a_in.C)
                step_records.append(("block", stage, start, a_in, cur))
        # ---- features output: fp32 NCHW (classifiers.py:109 consumes it), or, pooled, its global average [N][C]
        # (AdaptiveAvgPool2d(1) fused with the last BatchNorm+ReLU: the feature map is never written)
        if pooled:
            self.out_shape = (N, cur.C)
            j = fwd.add(L.OP_POOL_ACT, [N, cur.H * cur.W, cur.C], [], cur.act_ptrs() + [None])
            self.patch_out = (j, 3)
        else:
            self.out_shape = (N, cur.C, cur.H, cur.W)
            j = fwd.add(L.OP_ADD_ACT, [cur.C, cur.H * cur.W], [float(N * cur.H * cur.W)],
                        cur.act_ptrs() + [None, None, None, None, None])
            self.patch_out = (j, 7)
        self.fwd_ops, self.fwd_n = fwd.build()
        self.final = cur

        # ---- backward ------------------------------------------------------------------------------
        self.bwd_segments = []      # [(stage, ops, n)]
        self.patch_gout = None
        self.patch_dx = None
        self.patch_x_bwd = None
        self._graphs = {}
        # what the launch lists were BUILT with (the capture decision in _run must not follow later changes of the engine's
        # switches: a list with stream-1 fork/join pairs or live event records cannot be captured whatever the switches say now)
        self._built_side = bool(eng.use_side_stream)
        self._built_prof = bool(eng.profile_opcodes)
        self._out_buf = None
        self._gout_buf = None
        self._x_buf = None
        self._x_direct = None
        if training:
            self._build_backward(step_records, cur)

    def _build_backward(self, step_records, cur):
        """The backward launch lists, one per features.<stage> segment, from the forward's records (reverse order)."""
        eng, lib, N, H, W, need_dx, pooled, records = self.eng, self.eng.lib, self.N, self.H, self.W, self.need_dx, self.pooled, self._records
        self._seg_ops: Dict[int, _OpList] = {}
        self._order: List[int] = []
        seg_ops, order = self._seg_ops, self._order


        last_stage = step_records[-1][1]
        g_final = self._new((N, cur.H, cur.W, cur.C))
        if pooled:
            j = self._seg(last_stage).add(L.OP_POOL_BWD, [N, cur.H * cur.W, cur.C], [], [None, g_final.data_ptr()])
        else:
            j = self._seg(last_stage).add(L.OP_NCHW_TO_NHWC, [N, cur.C, cur.H * cur.W], [], [None, g_final.data_ptr()])
        self.patch_gout = (last_stage, j, 0)

        # ---- merged bookkeeping launches (mnas_bwd_post): the weight-gradient reductions of the main-stream kernel that just
        # ran ride in the SAME launch as the next layer's BatchNorm-backward finalize (and the second level of a two-level
        # reduction in the one after that), instead of 2-3 tiny dependent launches in every gap of the main stream
        merge = eng.merge_post
        self._pend = {"w1": None, "w2": None, "ops": None}
        self._rot = 0
        self._rot_bufs = [eng.scratch_wgrad2, eng.scratch_wgrad3, eng.scratch_wgrad4]









        self._masked_g = set()       # data_ptr of gradient tensors stored masked (dz) by their producer
        g = g_final
        g_red = 0            # number of fused-reduce partial columns already written for the layer g belongs to
        self.patch_x_bwd = None
        for si in range(len(step_records) - 1, -1, -1):
            kind, stage, start, a_in, a_out = step_records[si]
            ops = self._seg(stage)
            first = si == 0
            # what produced the step's INPUT: a ConvBlock (virtual act) or a block's residual sum / the network input
            if kind == "conv":
                rec = records[start]
                need = (not first) or need_dx
                if rec[2] is not None and rec[2].bn is None and si > 0 and step_records[si - 1][0] == "block":
                    # input is the previous block's materialised sum r: its gradient G feeds that block's project conv
                    prev_p = records[step_records[si - 1][2] + 2]
                    tgt = (prev_p[3].data, prev_p[3].bn)
                else:
                    tgt = self._target_of(rec[2])
                g, g_red = self._conv_bwd(ops, rec, g, None, need, g_red, tgt)
            else:
                re_, rd, rp = records[start], records[start + 1], records[start + 2]
                G = g                                   # grad wrt the block output (materialised sum)
                se_rec = self._se_records.get(start + 2)
                if se_rec is not None and se_rec[5]:
                    # excitation on load: the project conv's backward runs on the UNGATED activation in segment mode; its
                    # weight-gradient slabs give du and (gated) dW3 without a pass over gs / a2 (csrc/mnas_se.hip)
                    gs_, du_ = self._conv_bwd_se_proj(ops, rp, G, g_red, se_rec)
                    g2, c2 = self._se_bwd(ops, start + 2, gs_, du_)
                else:
                    g2, c2 = self._conv_bwd(ops, rp, G, None, True, g_red, self._target_of(rp[2]))
                    if se_rec is not None:
                        g2, c2 = self._se_bwd(ops, start + 2, g2)         # g2 becomes dL/d(activated depthwise output)
                g1, c1 = self._conv_bwd(ops, rd, g2, None, True, c2, self._target_of(rd[2]))
                need = (not first) or need_dx
                if need:
                    # expand dgrad (+ skip gradient fused in its epilogue) produces the gradient of the block INPUT:
                    # either a virtual activation (producer conv) or the previous block's sum (-> its project conv)
                    if a_in.bn is not None:
                        tgt = self._target_of(a_in)
                    elif si > 0 and step_records[si - 1][0] == "block":
                        prev_p = records[step_records[si - 1][2] + 2]
                        tgt = (prev_p[3].data, prev_p[3].bn)
                    else:
                        tgt = None
                    g, g_red = self._conv_bwd(ops, re_, g1, G, True, c1, tgt)
                else:
                    self._conv_bwd(ops, re_, g1, None, False, c1, None)
                    g, g_red = None, 0
        if need_dx and not self.x_is_image:
            Cin = self.in_channels
            j = self._seg(step_records[0][1]).add(L.OP_ADD_ACT, [Cin, H * W], [float(N * H * W)],
                                            [g.data_ptr(), None, None, None, None, None, None, None])
            self.patch_dx = (step_records[0][1], j, 7)
        self._flush_post()
        built = {}
        for st in order:
            # the main stream waits for the side stream's weight gradients at the end of backward, and at the end of the
            # stages somebody consumes right away (Trainer: the stage that completes gradient bucket 0); joining after every
            # stage cost 0.1 ms/step of main-stream waits with nobody looking at the gradients
            if eng.join_stages is None:
                need = st == order[-1] or eng.on_stage_done is not None
            else:
                need = st == order[-1] or st in eng.join_stages
            if eng.use_side_stream and need:
                seg_ops[st].join()
            built[st] = seg_ops[st].build()
        self.bwd_segments = [(st,) + built[st] for st in order]
        self._seg_index = {st: n for n, st in enumerate(order)}
        # all segments as ONE list (graph mode without a stage-done callback: one hipGraphLaunch per backward instead of one per
        # stage -- every graph boundary is ~9 us of idle GPU)
        n_all = sum(built[st][1] for st in order)
        self.bwd_all = (L.MnasOp * max(1, n_all))()
        self.bwd_all_n, self._seg_off, k = n_all, {}, 0
        for st in order:
            arr, n = built[st]
            self._seg_off[st] = k
            for j in range(n):
                C.memmove(C.byref(self.bwd_all, k * C.sizeof(L.MnasOp)), C.byref(arr, j * C.sizeof(L.MnasOp)), C.sizeof(L.MnasOp))
                k += 1
        if self.patch_x_bwd is not None:
            ops_obj, idx, slot = self.patch_x_bwd
            st = [s for s in order if seg_ops[s] is ops_obj][0]
            self.patch_x_bwd = (st, idx, slot)
        if self.patch_dx is not None and not isinstance(self.patch_dx[0], int):
            ops_obj, idx, slot = self.patch_dx
            self.patch_dx = ([s for s in order if seg_ops[s] is ops_obj][0], idx, slot)

    def _new(self, shape, dtype=torch.bfloat16):
        t = torch.empty(shape, dtype=dtype, device=self.eng.device)
        self.keep.append(t)
        return t

    def _bnbuf(self, C_):
        t = torch.zeros((L_BN_ROWS, C_), dtype=torch.float32, device=self.eng.device)
        self.keep.append(t)
        return t

    def _conv_fwd(self, ci: _ConvInfo, a_in: Optional[_Act], Hi, Wi):
        eng, lib, dev, N, H, W, training = self.eng, self.eng.lib, self.eng.device, self.N, self.H, self.W, self.training
        new, bnbuf, fwd, records = self._new, self._bnbuf, self._fwd, self._records
        Ho, Wo = ci.out_hw(Hi, Wi)
        M = N * Ho * Wo
        y = new((N, Ho, Wo, ci.cout))
        bn = bnbuf(ci.cout)
        conv, bnm = ci.mod.conv, ci.mod.bn
        bias = conv.bias.data_ptr() if conv.bias is not None else None
        nparts = lib.mnas_conv_gemm_parts(0, M, ci.cin, ci.cout, ci.k * ci.k) if ci.kind in ("pw", "dense") else -1
        if nparts < 1:
            nparts = max(1, min(eng.igemm_fwd_parts, _cdiv(M, 128 if M >= _SMALL_M else lib.mnas_conv_gemm_tile_pixels(M, ci.cout, ci.k * ci.k * ci.cin))))
        if ci.kind == "dense":       # small maps: one image per workgroup (csrc/mnas_dimg.hip)
            ip = lib.mnas_conv_img_parts(0, N, Hi, Wi, ci.cin, Ho, Wo, ci.cout, ci.k, ci.stride, ci.pad)
            nparts = ip if ip > 0 else nparts
        stats = eng.scratch_stats.data_ptr() if training else None
        if ci.kind == "stem":
            sp = lib.mnas_stem_parts(0, N, Hi, Wi, ci.cout)
            if self._aff_ptr is not None and (sp < 1 or (training and lib.mnas_stem_parts(1, N, Hi, Wi, ci.cout) < 1)):
                # the fused Normalize / uint8 load exists only in the band kernels (csrc/mnas_stem.hip)
                raise NotImplementedError(
                    "fused input normalisation needs the stem band kernels: 32 output channels and an image width that is a "
                    "multiple of 4 (16 in training); got %dx%d, %d channels.  Normalise on the host or call "
                    "set_input_normalization(None, None)" % (Hi, Wi, ci.cout))
            nparts = sp if sp > 0 else nparts
            j = fwd.add(L.OP_STEM_FWD, [N, Hi, Wi, Ho, Wo, ci.cout, nparts, 1 if self.in_u8 else 0], [],
                        [None, ci.w_fwd.data_ptr(), bias, y.data_ptr(), stats, self._aff_ptr])
            self.patch_x.append((j, 0))
        elif ci.kind == "dw":
            nlaunch = max(64, min(_STATS_PARTS, _cdiv(M * ci.cout, 256 * 16 * 2)))
            # stride 2 = SepConv(reduce=True)'s depthwise conv (mnasnet.py:73-81): plain direct kernels (csrc/mnas_dw2.hip)
            fwd.add(L.OP_DW_FWD, [N, Hi, Wi, ci.cout, ci.k, nlaunch, ci.stride], [],
                    a_in.act_ptrs() + [ci.w_fwd.data_ptr(), bias, y.data_ptr(), stats])
            nparts = lib.mnas_dw_rows(N, Hi, Wi, ci.cout, ci.k, nlaunch, 0 if ci.stride == 1 else 4)      # columns of the stats table
            if nparts < 1:
                raise RuntimeError("unsupported depthwise shape %s" % ((N, Hi, Wi, ci.cout, ci.k),))
        else:
            gate = [None, None, a_in.gate.data_ptr()] if a_in.gate is not None else []
            fwd.add(L.OP_CONV_GEMM, [0, N, Hi, Wi, ci.cin, Ho, Wo, ci.cout, ci.k, ci.k, ci.stride, ci.pad, nparts], [],
                    a_in.act_ptrs() + [None, None, None, ci.w_fwd.data_ptr(), bias, None, y.data_ptr(), stats] + gate)
        fwd.add(L.OP_BN_FWD_FINALIZE, [nparts, ci.cout, 1 if training else 0], [float(M), bnm.momentum, bnm.eps],
                [stats, bnm.weight.data_ptr(), bnm.bias.data_ptr(), bnm.running_mean.data_ptr(),
                 bnm.running_var.data_ptr(), bnm.num_batches_tracked.data_ptr(), bn.data_ptr()])
        out = _Act(y, bn, Ho, Wo, ci.cout)
        records.append(("conv", ci, a_in, out, Hi, Wi))
        return out

    def _se_onload_kseg(self, p_ci: _ConvInfo, h2: _Act, Hi, Wi):
        """Workgroups per image of the project conv's segment-mode backward when the excitation can be applied ON LOAD for this
        block (Engine.se_on_load): forward = MnasConvGemm.gate, backward = mnas_pw_bwd on the ungated activation with
        per-image weight-gradient slabs + mnas_se_proj_finalize.  0: keep the materialised a*s (k_se_scale) path."""
        eng, lib, N = self.eng, self.eng.lib, self.N
        HW, M, Ci, Co = Hi * Wi, self.N * Hi * Wi, p_ci.cin, p_ci.cout
        if not eng.se_on_load or h2.bn is None or p_ci.kind != "pw" or not lib.mnas_conv_gemm_gate_ok(N, HW, Ci, Co):
            return 0
        if not self.training:
            return 1
        if M < eng.pw_fused_min_pixels or not lib.mnas_pw_bwd_supported(Ci, Co):
            return 0
        tile, slices = lib.mnas_pw_bwd_tile_pixels(Ci, Co), lib.mnas_pw_bwd_slices(Ci, Co)
        return se_segments_per_image(N, HW, tile, slices, eng.scratch_wgrad2.numel() // (Co * Ci))

    def _se_fwd(self, se: _SEInfo, h2: _Act, Hi, Wi, p_ci: _ConvInfo):
        """squeeze-excite on the activated depthwise output (csrc/mnas_se.hip): pooled mean -> fc1+ReLU -> fc2 -> a2 * sigmoid.
        Returns what the project conv reads: the MATERIALISED scaled activation, or (Engine.se_on_load, supported shapes) the
        virtual activation with the excitation as a per-(image, channel) gate applied on load."""
        eng, lib, dev, N, H, W, training = self.eng, self.eng.lib, self.eng.device, self.N, self.H, self.W, self.training
        new, bnbuf, fwd, records = self._new, self._bnbuf, self._fwd, self._records
        E_, R_ = se.channels, se.reduced
        z = new((N, E_), torch.float32)
        hb = new((N, R_), torch.float32)
        u = new((N, E_), torch.float32)
        m_ = se.mod
        fwd.add(L.OP_POOL_ACT, [N, Hi * Wi, E_], [], h2.act_ptrs() + [z.data_ptr()])
        kseg = self._se_onload_kseg(p_ci, h2, Hi, Wi)
        gate = new((N, E_), torch.float32) if kseg else None
        if eng.se_fused_mlp and lib.mnas_se_fc_supported(E_, R_):
            # the whole excitation MLP (+ the gate table) in one launch (csrc/mnas_se.hip k_se_fc_fwd)
            fwd.add(L.OP_SE_FC_FWD, [N, E_, R_], [], [z.data_ptr(), m_.fc1.weight.data_ptr(), m_.fc1.bias.data_ptr(), m_.fc2.weight.data_ptr(),
                                                      m_.fc2.bias.data_ptr(), hb.data_ptr(), u.data_ptr(), gate.data_ptr() if kseg else None])
        else:
            fwd.add(L.OP_HEAD_LINEAR, [N, E_, R_, 1, 0, 0], [], [z.data_ptr(), m_.fc1.weight.data_ptr(), m_.fc1.bias.data_ptr(), hb.data_ptr()])
            fwd.add(L.OP_HEAD_LINEAR, [N, R_, E_, 0, 0, 0], [], [hb.data_ptr(), m_.fc2.weight.data_ptr(), m_.fc2.bias.data_ptr(), u.data_ptr()])
            if kseg:
                fwd.add(L.OP_SE_GATE, [N, E_], [], [u.data_ptr(), gate.data_ptr()])
        self._se_records[len(records)] = (se, h2, z, hb, u, kseg, gate)    # keyed by the record index of the project conv that follows
        if kseg:
            return _Act(h2.data, h2.bn, Hi, Wi, E_, gate)
        a2s = new((N, Hi, Wi, E_))
        fwd.add(L.OP_SE_SCALE, [N, Hi * Wi, E_], [], h2.act_ptrs() + [u.data_ptr(), a2s.data_ptr()])
        return _Act(a2s, None, Hi, Wi, E_)

    def _seg(self, stage):
        if stage not in self._seg_ops:
            self._seg_ops[stage] = _OpList(self.eng, "bwd")
            self._order.append(stage)
        return self._seg_ops[stage]

    def _next_scratch(self):
        b = self._rot_bufs[self._rot % 3]
        self._rot += 1
        return b

    def _emit_post(self, ops: _OpList, bn=None):
        pend = self._pend
        w1, w2 = pend["w1"], pend["w2"]
        pend["w1"] = pend["w2"] = None
        if w1 is not None and w1[7] == 2:
            pend["w2"] = w1[:7] + (3,)
        if bn is None and w1 is None and w2 is None:
            return
        none = (None, None, 0, 0, 0, 0, 0, 0)
        w1 = w1 or none
        w2 = w2 or none
        bn = bn or (None, None, None, None, 0, 0, 0.0)
        ops.add(L.OP_BWD_POST, [bn[4], bn[5]] + list(w1[2:]) + list(w2[2:]), [bn[6]],
                list(bn[:4]) + [w1[0], w1[1], w2[0], w2[1]], 0)
        pend["ops"] = ops

    def _flush_post(self):
        pend = self._pend
        while pend["w1"] is not None or pend["w2"] is not None:
            self._emit_post(pend["ops"])

    def _queue_wgrad(self, ops, partial, nsplit, Co_, Ci_, taps, dw, grad_ptr):
        pend = self._pend
        pend["w1"] = (partial, grad_ptr, nsplit, Co_, Ci_, taps, 1 if dw else 0, 1 if nsplit <= 256 else 2)
        pend["ops"] = ops

    def _conv_bwd(self, ops: _OpList, rec, g, resid, need_gin, g_reduced=False, red_target=None):
        """Backward of one ConvBlock application.  g: bf16 grad wrt its activated output.
        g_reduced: the producer of g already wrote this layer's BN-backward partial sums into
        eng.scratch_red (fused epilogue) with `g_reduced` columns.  red_target: (y, bn, C) of the ConvBlock
        whose activated output is THIS layer's input -- the dgrad epilogue then does that reduce.
        Returns (gin, ncols) : bf16 grad wrt the (activated) input or None, and the number of partial
        columns written for red_target (0 if not fused)."""
        eng, lib, N, merge, pend, records = self.eng, self.eng.lib, self.N, self.eng.merge_post, self._pend, self._records
        new = self._new
        _, ci, a_in, out, Hi, Wi = rec
        Ho, Wo, Co = out.H, out.W, ci.cout
        M = N * Ho * Wo
        gy = [g.data_ptr(), out.data.data_ptr(), out.bn.data_ptr()]
        if g_reduced:
            nred = g_reduced
            red_buf = eng.scratch_red
        else:
            nred = max(1, min(1024, _cdiv(M * Co, 256 * 8 * 8)))
            red_buf = eng.scratch_stats
            ops.add(L.OP_BN_BWD_REDUCE, [Co, nred], [float(M)], gy[:2] + [out.bn.data_ptr(), red_buf.data_ptr()])
        if merge:
            if pend["ops"] is not None and pend["ops"] is not ops:
                self._flush_post()               # a stage's gradients are complete inside its own launch list
            self._emit_post(ops, (red_buf.data_ptr(), out.bn.data_ptr(), eng.gptr(ci, 2), eng.gptr(ci, 3), nred, Co, float(M)))
        else:
            ops.add(L.OP_BN_BWD_FINALIZE, [nred, Co, 1], [float(M)],
                    [red_buf.data_ptr(), out.bn.data_ptr(), eng.gptr(ci, 2), eng.gptr(ci, 3)])
        # the weight-gradient kernels only share READ-ONLY inputs (g, y, the dy coefficients just finalised, the
        # forward activations) with the input-gradient chain: with Engine.use_side_stream they go to the side stream
        gyd = gy
        if ci.kind == "dense" and eng.materialize_dy:
            # dense 3x3: every dy element is gathered 2.25-10 times by the input/weight-gradient kernels; form it once
            dyb = new((N, Ho, Wo, Co))
            ops.add(L.OP_DY_MAT, [Co], [float(M)], gy + [dyb.data_ptr()], 0)
            gyd = [dyb.data_ptr(), None, None]
        WS = 1 if (eng.use_side_stream and M <= eng.side_stream_max_pixels) else 0
        if WS:
            ops.fork()
        gin, ncols = None, 0
        rt = red_target if (red_target is not None and need_gin) else None
        if ci.kind == "stem":
            nsp = max(1, min(512, _cdiv(M, 1024)))
            sp = lib.mnas_stem_parts(1, N, Hi, Wi, Co)
            nsp = min(sp if sp > 0 else nsp, _STEM_WGRAD_PARTS_MAX)
            if nsp * Co * 27 > eng.scratch_wgrad.numel():
                raise RuntimeError("stem weight-gradient scratch too small (%d splits)" % nsp)
            jx = ops.add(L.OP_STEM_WGRAD, [N, Hi, Wi, Ho, Wo, Co, nsp, 1 if self.in_u8 else 0], [],
                         [None] + gy + [eng.scratch_wgrad.data_ptr(), self._aff_ptr], WS)
            self.patch_x_bwd = (ops, jx, 0)
            ops.add(L.OP_WGRAD_FINALIZE, [nsp, Co, 27, 1, 1], [], [eng.scratch_wgrad.data_ptr(), eng.gptr(ci, 0)], WS)
            if need_gin:
                # dL/d image (fp32 NCHW; csrc/mnas_stem.hip k_stem_dgrad): not on the training path, autograd completeness only
                if self.in_u8:
                    raise RuntimeError("a uint8 image has no gradient")
                jd = ops.add(L.OP_STEM_DGRAD, [N, Hi, Wi, Ho, Wo, Co], [], gy + [ci.mod.conv.weight.data_ptr(), self._aff_ptr, None], 0)
                self.patch_dx = (ops, jd, 5)
        elif ci.kind == "dw" and ci.stride == 2:
            # SepConv(reduce=True): two plain launches, no fused reduce (the producer of x runs its own mnas_bn_bwd_reduce)
            nparts = max(64, min(eng.dw_bwd_parts, _cdiv(M * Co, 256 * 16 * 2)))
            if g.data_ptr() in self._masked_g:
                raise AssertionError("masked gradient handed to the stride-2 depthwise backward")
            wrows = lib.mnas_dw_rows(N, Hi, Wi, Co, ci.k, nparts, 7)
            if wrows < 1 or wrows * ci.k * ci.k * Co > eng.scratch_wgrad.numel():
                raise RuntimeError("unsupported stride-2 depthwise shape %s" % ((N, Hi, Wi, Co, ci.k),))
            dwp = a_in.act_ptrs() + gy + [ci.w_fwd.data_ptr(), None, eng.scratch_wgrad.data_ptr(), None, None]
            ops.add(L.OP_DW_BWD, [N, Hi, Wi, Co, ci.k, nparts, 2, 2, 0], [], dwp, WS)              # weight gradient
            ops.add(L.OP_DW_WGRAD_FINALIZE, [wrows, Co, ci.k, 1], [], [eng.scratch_wgrad.data_ptr(), eng.gptr(ci, 0)], WS)
            if need_gin:
                gin = new((N, Hi, Wi, ci.cin))
                dwp = a_in.act_ptrs() + gy + [ci.w_fwd.data_ptr(), gin.data_ptr(), None, None, None]
                ops.add(L.OP_DW_BWD, [N, Hi, Wi, Co, ci.k, nparts, 1, 2, 0], [], dwp, 0)           # input gradient
        elif ci.kind == "dw":
            nparts = max(64, min(eng.dw_bwd_parts, _cdiv(M * Co, 256 * 16 * 2)))
            gin = new((N, Hi, Wi, ci.cin))
            red = [None, None]
            if rt is not None:
                red = [rt[1].data_ptr(), eng.scratch_red.data_ptr()]
                ncols = lib.mnas_dw_rows(N, Hi, Wi, Co, ci.k, nparts, 1 if ci.k in eng.dw_fused_k else 2)
            wsc = (self._next_scratch() if merge else eng.scratch_wgrad2) if ci.k in eng.dw_fused_k else eng.scratch_wgrad     # fused: main stream
            dwp = a_in.act_ptrs() + gy + [ci.w_fwd.data_ptr(), gin.data_ptr(), wsc.data_ptr()] + red
            wrows = lib.mnas_dw_rows(N, Hi, Wi, Co, ci.k, nparts, 1 if ci.k in eng.dw_fused_k else 3)
            if wrows < 1 or (rt is not None and ncols < 1):
                raise RuntimeError("unsupported depthwise shape %s" % ((N, Hi, Wi, Co, ci.k),))
            if ci.k in eng.dw_fused_k:
                # one sweep: dgrad + wgrad (+ fused reduce); both partial tables have `wrows` rows
                if rt is not None:
                    ncols = wrows
                # g written by a project conv's out-stage backward as dz = g*[s*y+t>0] (see below): dy-on-read skips the mask
                gm = 1 if g.data_ptr() in self._masked_g else 0
                if gm and rt is None:
                    raise AssertionError("masked gradient handed to a depthwise backward without the fused reduce")
                ops.add(L.OP_DW_BWD, [N, Hi, Wi, Co, ci.k, nparts, 0, 0, gm], [], dwp, 0)
                if merge:
                    self._queue_wgrad(ops, wsc.data_ptr(), wrows, Co, 1, ci.k * ci.k, True, eng.gptr(ci, 0))
                else:
                    ops.add(L.OP_DW_WGRAD_FINALIZE, [wrows, Co, ci.k, 1], [], [eng.scratch_wgrad2.data_ptr(), eng.gptr(ci, 0)], 0)
            else:
                if g.data_ptr() in self._masked_g:
                    raise AssertionError("masked gradient handed to the two-launch depthwise backward")
                ops.add(L.OP_DW_BWD, [N, Hi, Wi, Co, ci.k, nparts, 2], [], dwp, WS)          # weight gradient
                ops.add(L.OP_DW_WGRAD_FINALIZE, [wrows, Co, ci.k, 1], [], [eng.scratch_wgrad.data_ptr(), eng.gptr(ci, 0)], WS)
                ops.add(L.OP_DW_BWD, [N, Hi, Wi, Co, ci.k, nparts, 1], [], dwp, 0)           # input gradient
        elif (ci.kind == "pw" and need_gin and M >= eng.pw_fused_min_pixels and lib.mnas_pw_bwd_supported(ci.cin, Co)
              and not (M < eng.pw_split_max_pixels and resid is None and Co < ci.cin and Co <= 128
                       and lib.mnas_conv_gemm_parts(1, M, Co, ci.cin, 1) > 0)):
            # large-pixel-count 1x1 conv: ONE sweep produces the input gradient, the weight-gradient partials and the
            # fused reduce (both former kernels stream the same g, y; see csrc/mnas_pwbwd.hip).  Main stream.
            gin = new((N, Hi, Wi, ci.cin))
            nparts = max(1, min(eng.pw_bwd_parts_large if M >= 800000 else (eng.pw_bwd_parts_mid if M >= 100000 else eng.pw_bwd_parts_small),
                                _cdiv(M, 128 if max(ci.cin, Co) <= 80 else 64)))
            red = [None, None, None]
            if rt is not None:
                red = [eng.scratch_red.data_ptr(), rt[0].data_ptr(), rt[1].data_ptr()]
                ncols = nparts
            wsc = self._next_scratch() if merge else eng.scratch_wgrad2
            gyp, extra = gy, [None, None, None]
            if eng.pw_recompute_y and (lib.mnas_pw_bwd_forms(ci.cin, Co) & 2):
                # widening (expand) conv: dy-on-load's raw forward output is recomputed from the staged x tile on the matrix
                # cores (bit-identical to the stored tensor) instead of being read: a third of the launch's reads
                conv = ci.mod.conv
                gyp = [gy[0], None, gy[2]]
                extra = [None, ci.w_fwd.data_ptr(), conv.bias.data_ptr() if conv.bias is not None else None]
            # project conv in front of a depthwise conv: the out-stage form stores the input gradient already masked with the
            # depthwise conv's ReLU (the mask its fused reduce computes anyway); mnas_dw_bwd then skips re-deriving it per window column
            masked = 0
            if (eng.dw_masked_g and rt is not None and resid is None and a_in.bn is not None and rt[0] is a_in.data
                    and (lib.mnas_pw_bwd_forms(ci.cin, Co) & 4) and self._feeds_fused_dw(a_in)):
                masked = 1
                self._masked_g.add(gin.data_ptr())
            seg = 0
            if eng.pw_bwd_segments and M >= 800000 and (lib.mnas_pw_bwd_forms(ci.cin, Co) & 4):
                # the out-stage (project) convs of the 112x112 / 56x56 stages: contiguous pixel range per workgroup (segment mode of
                # csrc/mnas_pwbwd.hip) and one resident round of workgroups instead of tiles strided over a 1024-wide grid:
                # 48 -> 16 at 112x112 200 -> 189 us, 72 -> 24 at 56x56 87 -> 83 us (same call); the expand forms (3-4 resident
                # workgroups per CU) and 240 -> 40 at 28x28 lose with it (16 -> 48: 155 -> 220 us at 512 segments)
                tile = lib.mnas_pw_bwd_tile_pixels(ci.cin, Co)
                nparts = max(1, min(nparts, eng.pw_bwd_segments))
                seg = _cdiv(_cdiv(M, nparts), tile) * tile
                nparts = _cdiv(M, seg)
                if rt is not None:
                    ncols = nparts
            ops.add(L.OP_PW_BWD, [M, ci.cin, Co, nparts, masked, seg], [],
                    a_in.act_ptrs() + gyp + [ci.w_dgrad.data_ptr(), resid.data_ptr() if resid is not None else None,
                                             gin.data_ptr(), wsc.data_ptr()] + red + extra, 0)
            if merge:
                self._queue_wgrad(ops, wsc.data_ptr(), nparts, Co, ci.cin, 1, False, eng.gptr(ci, 0))
            else:
                ops.add(L.OP_WGRAD_FINALIZE, [nparts, Co, ci.cin, 1, 1], [], [eng.scratch_wgrad2.data_ptr(), eng.gptr(ci, 0)], 0)
        else:
            # pixel splits: as many as keep slabs x splits within the workgroup budget (rounding UP put 513-540 workgroups on the
            # 512 resident slots of most launches: a second, nearly empty round)
            slabs = lib.mnas_conv_wgrad_slabs(Co, ci.cin, ci.k * ci.k)
            nsp = max(1, min(eng.wgrad_wgs // slabs, _cdiv(M, 256)))
            # partial[nsp][Co][K] must fit the scratch _setup sized for 1024 workgroups (Engine.wgrad_wgs is public)
            nsp = max(1, min(nsp, eng.scratch_wgrad.numel() // (Co * ci.cin * ci.k * ci.k)))
            ops.add(L.OP_CONV_WGRAD, [N, Hi, Wi, ci.cin, Ho, Wo, Co, ci.k, ci.k, ci.stride, ci.pad, nsp], [],
                    a_in.act_ptrs() + gyd + [eng.scratch_wgrad.data_ptr()], WS)
            ops.add(L.OP_WGRAD_FINALIZE, [nsp, Co, ci.cin, ci.k * ci.k, 1], [], [eng.scratch_wgrad.data_ptr(), eng.gptr(ci, 0)], WS)
            if need_gin:
                gin = new((N, Hi, Wi, ci.cin))
                Min = N * Hi * Wi
                nparts = lib.mnas_conv_gemm_parts(1, Min, Co, ci.cin, ci.k * ci.k)
                if nparts < 1:
                    nparts = max(1, min(eng.igemm_dgrad_parts, _cdiv(Min, 128 if Min >= _SMALL_M else lib.mnas_conv_gemm_tile_pixels(Min, ci.cin, ci.k * ci.k * Co))))
                tconv = (eng.use_tconv and ci.kind == "dense" and self._tconv_ok.get(id(ci), False) and gyd is not gy and resid is None
                         and Hi == 2 * Ho and Wi == 2 * Wo)
                if tconv:
                    tp = lib.mnas_tconv_parts(N, Ho, Wo, Co, ci.cin)       # (-1: a form that needs a larger batch)
                    tconv = tp > 0
                    nparts = tp if tconv else nparts
                if tconv:
                    pass
                elif ci.kind == "dense" and gyd is not gy and resid is None:
                    ip = lib.mnas_conv_img_parts(1, N, Ho, Wo, Co, Hi, Wi, ci.cin, ci.k, ci.stride, ci.pad)
                    nparts = ip if ip > 0 else nparts
                red = [None, None, None]
                if rt is not None:
                    red = [eng.scratch_red.data_ptr(), rt[0].data_ptr(), rt[1].data_ptr()]
                    ncols = nparts
                if tconv:
                    # stride-2 3x3: transposed convolution over the materialised dy (csrc/mnas_tconv.hip)
                    ops.add(L.OP_TCONV_DGRAD, [N, Ho, Wo, Co, ci.cin, nparts], [],
                            [gyd[0], ci.w_tconv.data_ptr(), gin.data_ptr(), red[0], red[1], red[2]])
                else:
                    ops.add(L.OP_CONV_GEMM, [1, N, Ho, Wo, Co, Hi, Wi, ci.cin, ci.k, ci.k, ci.stride, ci.pad, nparts], [],
                            [None, None, None] + gyd + [ci.w_dgrad.data_ptr(), None,
                                                       resid.data_ptr() if resid is not None else None,
                                                       gin.data_ptr(), red[0], red[1], red[2]])
        if ci.kind == "dw" and resid is not None:
            raise AssertionError("residual add into a depthwise dgrad does not occur")
        return gin, ncols

    def _feeds_fused_dw(self, act: _Act):
        """True if the ConvBlock that produced the virtual activation `act` is a depthwise conv whose backward runs as the fused
        sweep (the only mnas_dw_bwd form that takes a masked gradient)."""
        for rec in self._records:
            if rec[3] is act or (rec[3].data is act.data and rec[3].bn is act.bn):
                ci = rec[1]
                # ... and carries the fused reduce (its own input is a virtual activation), the form g_masked exists for
                return ci.kind == "dw" and ci.stride == 1 and ci.k in self.eng.dw_fused_k and rec[2] is not None and rec[2].bn is not None
        return False

    def _conv_bwd_se_proj(self, ops: _OpList, rec, g, g_reduced, se_rec):
        """Backward of the project conv of a squeeze-excite block whose excitation is applied on load.  Returns (gs, du): the input
        gradient wrt the GATED activation and dL/du (fp32 [N][E])."""
        eng, lib, N, merge = self.eng, self.eng.lib, self.N, self.eng.merge_post
        new = self._new
        _, ci, a_in, out, Hi, Wi = rec
        se, h2, z, hb, u, kseg = se_rec[:6]
        Co, M, HW = ci.cout, N * Hi * Wi, Hi * Wi
        gy = [g.data_ptr(), out.data.data_ptr(), out.bn.data_ptr()]
        if g_reduced:
            nred, red_buf = g_reduced, eng.scratch_red
        else:
            nred = max(1, min(1024, _cdiv(M * Co, 256 * 8 * 8)))
            red_buf = eng.scratch_stats
            ops.add(L.OP_BN_BWD_REDUCE, [Co, nred], [float(M)], gy[:2] + [out.bn.data_ptr(), red_buf.data_ptr()])
        if merge:
            if self._pend["ops"] is not None and self._pend["ops"] is not ops:
                self._flush_post()
            self._emit_post(ops, (red_buf.data_ptr(), out.bn.data_ptr(), eng.gptr(ci, 2), eng.gptr(ci, 3), nred, Co, float(M)))
        else:
            ops.add(L.OP_BN_BWD_FINALIZE, [nred, Co, 1], [float(M)],
                    [red_buf.data_ptr(), out.bn.data_ptr(), eng.gptr(ci, 2), eng.gptr(ci, 3)])
        gs = new((N, Hi, Wi, ci.cin))
        du = new((N, ci.cin), torch.float32)
        wsc = self._next_scratch() if merge else eng.scratch_wgrad2
        ops.add(L.OP_PW_BWD, [M, ci.cin, Co, N * kseg, 0, HW // kseg], [],
                h2.act_ptrs() + gy + [ci.w_dgrad.data_ptr(), None, gs.data_ptr(), wsc.data_ptr()] + [None] * 6, 0)
        ops.add(L.OP_SE_PROJ_FIN, [N, kseg, Co, ci.cin, 1], [],
                [wsc.data_ptr(), u.data_ptr(), ci.mod.conv.weight.data_ptr(), eng.gptr(ci, 0), du.data_ptr()], 0)
        return gs, du

    def _se_bwd(self, ops: _OpList, rec_index, gs, du=None):
        """Backward of the squeeze-excite stage: gs = dL/d(a2 * s) from the project conv's input gradient -> dL/d a2, and the
        SE parameters' gradients (accumulated into the flat buffer; shared blocks sum their applications).  du: dL/du when the
        project conv's backward already produced it (excitation on load), else it is reduced here from (gs, a2)."""
        eng, lib, N, merge, pend, records = self.eng, self.eng.lib, self.N, self.eng.merge_post, self._pend, self._records
        new = self._new
        se, h2, z, hb, u = self._se_records[rec_index][:5]
        E_, R_ = se.channels, se.reduced
        HWl = h2.H * h2.W
        m_ = se.mod
        dh = new((N, R_), torch.float32)
        dzp = new((N, E_), torch.float32)
        ga = new((N, h2.H, h2.W, E_))
        if du is None:
            du = new((N, E_), torch.float32)
            sb = lib.mnas_se_scratch_bytes(N, HWl, E_)
            if sb < 0:
                raise RuntimeError("unsupported squeeze-excite shape %s" % ((N, HWl, E_),))
            dup = new((sb // 4,), torch.float32)
            ops.add(L.OP_SE_BWD_REDUCE, [N, HWl, E_], [], [gs.data_ptr()] + h2.act_ptrs() + [u.data_ptr(), du.data_ptr(), dup.data_ptr()], 0)
        if eng.se_fused_mlp and lib.mnas_se_fc_supported(E_, R_):
            # the MLP backward in one op (two kernels: per-image dh / dz, then the parameter gradients; csrc/mnas_se.hip)
            ops.add(L.OP_SE_FC_BWD, [N, E_, R_, 1], [], [du.data_ptr(), z.data_ptr(), hb.data_ptr(), m_.fc1.weight.data_ptr(),
                                                         m_.fc2.weight.data_ptr(), dh.data_ptr(), dzp.data_ptr(), eng.gptr(se, 0),
                                                         eng.gptr(se, 1), eng.gptr(se, 2), eng.gptr(se, 3)], 0)
        else:
            self._se_mlp_bwd_unfused(ops, se, z, hb, du, dh, dzp)
        # the BatchNorm2-backward reduce of the depthwise conv rides in the same pass (ga is its g; h2 = its raw output + bnbuf)
        ncols = lib.mnas_se_bwd_apply_cols(N, HWl, E_)
        fused = h2.bn is not None and 0 < ncols <= _STATS_PARTS
        ops.add(L.OP_SE_BWD_APPLY, [N, HWl, E_], [],
                [gs.data_ptr(), u.data_ptr(), dzp.data_ptr(), ga.data_ptr()] +
                ([h2.data.data_ptr(), h2.bn.data_ptr(), eng.scratch_red.data_ptr()] if fused else [None, None, None]), 0)
        return ga, (ncols if fused else 0)

    def _se_mlp_bwd_unfused(self, ops, se, z, hb, du, dh, dzp):
        """The excitation MLP's backward as four mnas_head_linear_* launches (Engine.se_fused_mlp = False: the A/B baseline)."""
        eng, N, m_ = self.eng, self.N, se.mod
        E_, R_ = se.channels, se.reduced
        # fc2: dW2 += du^T hb, db2 += sum du ; dh = (du W2) * [hb > 0]
        ops.add(L.OP_HEAD_LINEAR, [N, R_, E_, 0, 1, 1], [], [hb.data_ptr(), m_.fc2.weight.data_ptr(), None, None, du.data_ptr(),
                                                            eng.gptr(se, 2), eng.gptr(se, 3)], 0)
        ops.add(L.OP_HEAD_LINEAR, [N, R_, E_, 0, 0, 2], [], [hb.data_ptr(), m_.fc2.weight.data_ptr(), None, None, du.data_ptr(),
                                                            None, None, dh.data_ptr(), hb.data_ptr()], 0)
        # fc1: dW1 += dh^T z, db1 += sum dh ; dz = dh W1
        ops.add(L.OP_HEAD_LINEAR, [N, E_, R_, 1, 1, 1], [], [z.data_ptr(), m_.fc1.weight.data_ptr(), None, None, dh.data_ptr(),
                                                            eng.gptr(se, 0), eng.gptr(se, 1)], 0)
        ops.add(L.OP_HEAD_LINEAR, [N, E_, R_, 1, 0, 2], [], [z.data_ptr(), m_.fc1.weight.data_ptr(), None, None, dh.data_ptr(),
                                                            None, None, dzp.data_ptr(), None], 0)

    @staticmethod
    def _target_of(act: Optional[_Act]):
        """(raw y tensor, bnbuf) of the ConvBlock that produced a VIRTUAL activation, else None."""
        if act is None or act.bn is None:
            return None
        return (act.data, act.bn)


    # ------------------------------------------------------------------------------------------
    def _run(self, arr, n, what):
        failed = C.c_int(-1)
        streams = (C.c_void_p * 2)(L.cur_stream(), self.eng.side_stream_handle())
        live_events = self._built_prof and self.eng.profile_gate.value != 0      # bracketing event records: per-launch path
        # (not with the second stream: a segment may fork onto it and join only at the end of a later segment -- an unjoined capture)
        if self.eng.use_graphs and n > 1 and not live_events and not self._built_side:
            # the list as a hipGraph (csrc/mnas_abi.hip mnas_graph_create): captured at first use and whenever a run-time
            # pointer of the list (input batch, output, incoming gradient) or the event-record gate has changed since
            key = hash(bytes(arr))
            slot = self._graphs.setdefault(id(arr), {"exec": None, "key": None, "miss": 0})
            if slot["key"] != key and slot["miss"] < 4:
                if slot["exec"] is not None:
                    # the previous launch of this exec may still be running (the host enqueues a step in < 1 ms of a 10 ms step)
                    # and the exec owns its kernel arguments: wait for the device first (at most 4 re-captures per list)
                    torch.cuda.synchronize(self.eng.device)
                    self.eng.lib.mnas_graph_destroy(slot["exec"])
                    slot["exec"] = None
                    slot["miss"] += 1             # a list whose pointers change every step is not worth capturing: fall through
                ex = C.c_void_p()
                # capture on a stream of our own (the legacy default stream cannot capture); the graph is LAUNCHED on the current one
                if self.eng._capture_stream is None or self.eng._capture_stream.device != self.eng.device:
                    self.eng._capture_stream = torch.cuda.Stream(device=self.eng.device)
                cap = (C.c_void_p * 2)(self.eng._capture_stream.cuda_stream, streams[1])
                rc = self.eng.lib.mnas_graph_create(arr, n, cap, 2, C.byref(ex), C.byref(failed))
                if rc != 0:
                    raise RuntimeError("%s: graph capture failed with code %d at op %d" % (what, rc, failed.value))
                slot["exec"], slot["key"] = ex, key
            if slot["key"] == key and slot["exec"] is not None:
                rc = self.eng.lib.mnas_graph_launch(slot["exec"], streams[0])
                if rc != 0:
                    raise RuntimeError("%s: hipGraphLaunch failed with code %d" % (what, rc))
                return
        rc = self.eng.lib.mnas_run_ops_multi(arr, n, streams, 2, C.byref(failed))
        if rc != 0:
            raise RuntimeError("%s: mnas_run_ops failed with code %d at op %d (opcode %d)" %
                               (what, rc, failed.value, arr[failed.value].opcode if failed.value >= 0 else -1))

    def destroy_graphs(self):
        """Destroy the captured graph executables -- after a device sync: one of them may still be executing."""
        slots = [s for s in getattr(self, "_graphs", {}).values() if s["exec"] is not None]
        if not slots:
            return
        torch.cuda.synchronize(self.eng.device)
        for slot in slots:
            self.eng.lib.mnas_graph_destroy(slot["exec"])
            slot["exec"] = None

    def __del__(self):
        try:
            self.destroy_graphs()
        except Exception:
            pass

    def run_forward(self, x, static_io=False):
        """static_io (Trainer's autograd-free path with Engine.use_graphs): the output lives in a buffer of the program, so that the
        captured graph's pointers stay valid from step to step (the caller consumes it before the next forward of this program)."""
        if static_io and self.eng.use_graphs:
            if self._out_buf is None or self._out_buf.device != x.device:
                self._out_buf = torch.empty(self.out_shape, dtype=torch.float32, device=x.device)
            out = self._out_buf
            # the input: a batch that is the same tensor step after step (a resident benchmark batch) is read in place; as soon as
            # a different tensor arrives (a data loader), every batch is copied into a buffer of the program first (fp32 bs 256:
            # 154 MB, ~35 us; uint8: a quarter) so that the captured graphs stay valid
            if self._x_direct is None:
                self._x_direct = x.data_ptr()
            if self._x_buf is not None or x.data_ptr() != self._x_direct:
                if self._x_buf is None or self._x_buf.shape != x.shape or self._x_buf.dtype != x.dtype or self._x_buf.device != x.device:
                    self._x_buf = torch.empty_like(x)
                self._x_buf.copy_(x)
                x = self._x_buf
        else:
            out = torch.empty(self.out_shape, dtype=torch.float32, device=x.device)
        for j, slot in self.patch_x:
            self.fwd_ops[j].p[slot] = x.data_ptr()
        j, slot = self.patch_out
        self.fwd_ops[j].p[slot] = out.data_ptr()
        self.x_ref = x
        self._run(self.fwd_ops, self.fwd_n, "forward")
        return out

    def run_backward(self, gout, on_stage_done: Optional[Callable[[int], None]] = None, static_io=False):
        if static_io and self.eng.use_graphs:
            if self._gout_buf is None or self._gout_buf.shape != gout.shape or self._gout_buf.device != gout.device:
                self._gout_buf = torch.empty_like(gout)
            self._gout_buf.copy_(gout)                       # (a few hundred KB) keeps the captured pointer valid
            gout = self._gout_buf
        segs = {st: (arr, n) for st, arr, n in self.bwd_segments}

        def patch(where, ptr):
            st, j, slot = where
            segs[st][0][j].p[slot] = ptr
            self.bwd_all[self._seg_off[st] + j].p[slot] = ptr

        patch(self.patch_gout, gout.data_ptr())
        if self.patch_x_bwd is not None:
            patch(self.patch_x_bwd, self.x_ref.data_ptr())
        dx = None
        if self.patch_dx is not None:
            dx = torch.empty((self.N, self.in_channels, self.H, self.W), dtype=torch.float32, device=gout.device)
            patch(self.patch_dx, dx.data_ptr())
        if self.eng.use_graphs and on_stage_done is None and not self._built_side:
            self._run(self.bwd_all, self.bwd_all_n, "backward")
            self.x_ref = None
            return dx
        for st, arr, n in self.bwd_segments:
            self._run(arr, n, "backward[stage %d]" % st)
            if on_stage_done is not None:
                on_stage_done(st)
        self.x_ref = None
        return dx


L_BN_ROWS = 8
_MAX_PROGRAMS_PER_SHAPE = 4     # forwards kept alive simultaneously per (N,H,W,mode): beyond this the caller is leaking graphs


class _Lease:
    """Ties a Program's activation buffers to the autograd graph of ONE forward.  The program is handed back when
    backward has consumed it OR when the graph is dropped without a backward (exception between forward and backward,
    LR finder, BatchNorm recalibration pass, ...): ``ctx`` dies with the graph and takes this object with it."""
    __slots__ = ("prog", "consumed", "__weakref__")

    def __init__(self, prog):
        self.prog, self.consumed = prog, False
        prog.busy = True

    def release(self):
        if self.prog is not None:
            self.prog.busy = False
            self.prog = None

    def __del__(self):
        self.release()


class _EngineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, track, pooled, x, *params):
        need_dx = track and x.requires_grad
        training = eng.root.training
        prog = eng.program(x.shape[0], x.shape[2], x.shape[3], training, need_dx, pooled, x.dtype == torch.uint8)
        out = prog.run_forward(x)
        ctx.eng = eng
        ctx.lease = _Lease(prog) if (training and track) else None
        return out

    @staticmethod
    def backward(ctx, gout):
        lease, eng = ctx.lease, ctx.eng
        if lease is None:
            raise RuntimeError("backward through an eval-mode / no-grad engine forward")
        if lease.consumed or lease.prog is None:
            # the activation buffers belong to a per-shape program that later forwards reuse: a second backward
            # (retain_graph=True) would read overwritten activations
            raise RuntimeError("the HIP engine keeps activations for ONE backward per forward; run the forward again "
                               "(retain_graph=True is not supported through mnasnet_pytorch_amd modules)")
        prog = lease.prog
        try:
            accumulate = eng.prepare_grads()
            dx = prog.run_backward(gout.contiguous().float(), eng.on_stage_done)
            eng.finish_grads(accumulate)
        finally:
            lease.consumed = True
            lease.release()
        return (None, None, None, dx) + (None,) * len(eng.params)


class Engine:
    """Owns the packed weights, scratch space, flat gradient buffer and per-shape programs of one module
    subtree."""

    def __init__(self, root: nn.Module):
        self.root = root
        self._se_mods = {}
        self.steps = _trace(root, self._se_mods)
        self.lib = None
        self.device = None
        self.programs: Dict[tuple, List[Program]] = {}
        self.on_stage_done: Optional[Callable[[int], None]] = None
        # unique ConvBlocks in first-use order
        self.info: Dict[int, _ConvInfo] = {}
        self.convs: List[_ConvInfo] = []
        for op, m, stage in self.steps:
            for cb in ([m] if op == "conv" else m):
                if id(cb) not in self.info:
                    ci = _ConvInfo(cb, stage)
                    self.info[id(cb)] = ci
                    self.convs.append(ci)
        for n, (op, m, stage) in enumerate(self.steps):
            for cb in ([m] if op == "conv" else m):
                if self.info[id(cb)].kind == "stem" and n != 0:
                    raise NotImplementedError("a 3-channel stride-2 conv is only supported as the first layer")
        # squeeze-excite modules (SE variant only): one _SEInfo per unique module, keyed by its block's expand ConvBlock
        self.se_info: Dict[int, _SEInfo] = {}
        self.ses: List[_SEInfo] = []
        by_mod = {}
        for op, m, stage in self.steps:
            if op == "block" and id(m[0]) in self._se_mods:
                se = self._se_mods[id(m[0])]
                if id(se) not in by_mod:
                    by_mod[id(se)] = _SEInfo(se, stage)
                    self.ses.append(by_mod[id(se)])
                self.se_info[id(m[0])] = by_mod[id(se)]
        self.param_owners = self.convs + self.ses      # everything that owns parameters, in the flat-gradient order below
        # flat gradient layout: later stages first (their gradients are complete first in backward)
        self.params: List[nn.Parameter] = []
        off = 0
        self.stage_ranges: Dict[int, List[int]] = {}
        for ci in sorted(self.param_owners, key=lambda c: -c.stage):
            a = off
            for j, p in enumerate(ci.params):
                ci.gslice[j] = (off, p.numel())
                off += p.numel()
                self.params.append(p)
            r = self.stage_ranges.setdefault(ci.stage, [a, off])
            r[0], r[1] = min(r[0], a), max(r[1], off)
        self.grad_numel = off
        self._sig = None
        self._ext_grad: Optional[torch.Tensor] = None
        # weight-gradient kernels on a second HIP stream, concurrent with the input-gradient chain.  +14 % when introduced (round 1);
        # by round 3 the main-stream kernels and k_wgrad_t fill the chip on their own and the overlap only trades time between the
        # two streams (11.17-11.21 ms with it, 11.07-11.13 without, same call) -- and it hid per-kernel gains on the main stream
        # (the K-streaming input gradient: neutral with the side stream, -0.1 ms without).  Off by default; the path stays tested.
        self.use_graphs = False          # launch lists replayed as hipGraphs (Program._run)
        self._capture_stream = None
        self.use_side_stream = False
        self.side_stream_max_pixels = 1 << 40     # with use_side_stream: only layers with at most this many output pixels fork
        # workgroups of a k_wgrad launch (pixel splits x 64x64 slabs): 512 rather than 1024 leaves the main stream's persistent
        # grids more of the chip while it runs beside them (11.57 vs 11.66 ms/step, three same-call A/B pairs; 256: 11.77)
        self.wgrad_wgs = 512
        self.join_stages = None          # stages after which the main stream joins the side stream (None: see Program)
        self.use_tconv = True            # stride-2 dense 3x3 input gradient as a transposed convolution (csrc/mnas_tconv.hip)
        self.merge_post = True           # BatchNorm-backward finalize + weight-gradient reductions of the main stream in one launch
        self.materialize_dy = True       # dense 3x3 convs: dy formed once (mnas_dy_materialize), gathered plain by dgrad / wgrad
        # depthwise kernel sizes whose backward runs as ONE fused sweep (input gradient + weight gradient + reduce).  5x5 too:
        # with 2-row DMA groups the fused form gets full-width strips and beats the two launches (155 vs 210 us at 56x56)
        # although it needs all 256 VGPRs; (3,) selects the split form (input gradient on main, weight gradient on side)
        self.dw_fused_k = (3, 5)
        # 1x1 convs with at least this many pixels use the fused backward (mnas_pw_bwd) when the shape is supported: measured
        # per launch at bs 256 against the dgrad + wgrad pair: 201 vs 399 us (16->48 @112^2), 233 vs 331 (48->16), 119 vs 285
        # (32->16), 90 vs 199 (24->72 @56^2), 105 vs 195 (72->24), 81 vs 141 (40->240 @28^2), 111 vs 175 (240->40)
        self.pw_fused_min_pixels = 50000
        # narrowing (project) 1x1 convs below this pixel count: DMA-pipelined input gradient on the main stream
        # (csrc/mnas_pwf.hip MODE 1) + weight gradient on the side stream, instead of the fused sweep.  Off: alone the input
        # gradient takes 68 us against the fused kernel's 116 (576->96 at 14x14, bs 256), but next to the side stream's
        # k_wgrad it takes 130 us and the step is 0.1 ms slower (12.56 vs 12.44 ms)
        self.pw_split_max_pixels = 0
        # expand convs of the 112x112 / 56x56 stages (16->48, 24->72): the fused 1x1 backward recomputes the conv's raw output
        # from its (narrow) input instead of reading the t-times wider stored tensor (mnas_pw_bwd RECOMP; round 4:
        # 209 -> 155 us per launch at 112x112, bit-identical results)
        self.pw_recompute_y = True
        # project convs' fused backward (out-stage forms) store the depthwise conv's incoming gradient already masked with its ReLU
        # (dz = g*[s*y+t>0], the mask the fused reduce computes anyway); the depthwise sweep's dy-on-read then skips the mask:
        # 40 of the 5x5 row body's ~530 vector instructions (round 4; results bit-identical)
        self.dw_masked_g = True
        self.se_on_load = True           # squeeze-excite excitation applied in the project conv's load (forward) / folded into its
                                         # weight-gradient slabs (backward) where the kernels support the shape; False: k_se_scale
        self.se_fused_mlp = True         # the squeeze-excite MLP as mnas_se_fc_fwd / mnas_se_fc_bwd (1 + 2 kernels per block instead of 3 + 4)
        self.pw_bwd_segments = 512       # > 0: the project convs' fused backward at >= 800 k pixels walks contiguous pixel segments,
                                         # at most this many workgroups (0: tiles strided over the grid everywhere)
        self.igemm_fwd_parts = 1024      # upper bound on the persistent pixel-workgroups of a k_igemm forward / input-gradient launch
        self.igemm_dgrad_parts = 1024
        self.dw_bwd_parts = 2048         # upper bound on the persistent workgroups of a depthwise backward launch (round 6: 1024 made
                                         # the 1280-item launches of the 14x14 / 7x7 stages walk 1.25 items per workgroup on 1020
                                         # workgroups; one item per workgroup: step 10.15 vs 10.19 ms, three interleaved pairs)
        self.pw_bwd_parts_large = 1024   # ... on the 112x112 / 56x56 stages
        self.pw_bwd_parts_mid = 512      # ... on the 28x28 stage
        self.pw_bwd_parts_small = 80     # persistent pixel-workgroups of the fused 1x1 backward on the 14x14 stage (x 6 channel slices): a
                                         # multiple of 8 (the slices of one pixel column share an XCD's L2) with 6 x 80 <= the 512 resident
                                         # slots; 85 -> 80: class -0.06 ms, 88 (528 workgroups, a second round): +0.15 ms
        self.side_stream = None
        self._in_norm, self._in_aff = None, {}
        self.profile_opcodes = None      # set of opcodes to bracket with HIP events (bench.py roofline leg)
        self.profile_filter = None       # optional predicate (opcode, ints) -> bool narrowing the bracketed launches
        self.profile_events = []         # [(tag, start_handle, stop_handle)]
        self.profile_gate = C.c_int(1)   # 0: the bracketing event records of the compiled programs are skipped (read at run time by
                                         # mnas_run_ops: bench.py switches them on for the last step of a timed window only)
        self._events = []                # every HIP event handle the compiled programs own (destroyed with them)
        first = self.steps[0][1] if self.steps[0][0] == "conv" else self.steps[0][1][0]
        self.in_channels_hint = self.info[id(first)].cin
        self.starts_with_stem = self.info[id(first)].kind == "stem"

    # ---- device state ---------------------------------------------------------------------------
    def _signature(self):
        sig = [p.data_ptr() for p in self.params]
        for ci in self.convs:
            bn = ci.mod.bn
            sig += [bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr()]
        return tuple(sig)

    def _setup(self, device):
        self.lib = L.load()
        if self.device is not None:
            self.reset_programs()
        self.device = device
        self._validate_modules()
        if self.side_stream is not None and self.side_stream.device != device:
            self.side_stream = None          # re-created lazily on the new device (side_stream_handle)
        nbytes = self.lib.mnas_packed_bytes
        smax, wmax = 0, 0
        for ci in self.convs:
            if ci.kind in ("pw", "dense"):
                ci.w_fwd = torch.empty(nbytes(L.PACK_FWD, ci.cout, ci.cin, ci.k, ci.k), dtype=torch.uint8, device=device)
                ci.w_dgrad = torch.empty(nbytes(L.PACK_DGRAD, ci.cout, ci.cin, ci.k, ci.k), dtype=torch.uint8, device=device)
                ci.w_tconv = None
                # transposed-conv form of the stride-2 input gradient: the buffer exists for every stride-2 dense conv (< 1 MB each);
                # whether a PROGRAM packs and uses it is decided from that program's real dy plane (Program.__init__)
                if ci.kind == "dense" and ci.stride == 2 and self.use_tconv and self.materialize_dy:
                    ci.w_tconv = torch.empty(nbytes(L.PACK_TCONV, ci.cout, ci.cin, 3, 3), dtype=torch.uint8, device=device)
                K = ci.k * ci.k * ci.cin
                slabs = self.lib.mnas_conv_wgrad_slabs(ci.cout, ci.cin, ci.k * ci.k)
                wmax = max(wmax, max(1, _cdiv(1024, slabs)) * ci.cout * K)
                if ci.kind == "pw" and self.lib.mnas_pw_bwd_supported(ci.cin, ci.cout):
                    wmax = max(wmax, 1024 * ci.cout * ci.cin)      # one slab per workgroup of the fused 1x1 backward
                if ci.kind == "pw":
                    wmax = max(wmax, 64 * ci.cout * ci.cin)        # fused block backward: one slab per image group (<= 64)
            elif ci.kind == "dw":
                ci.w_fwd = torch.empty(nbytes(L.PACK_DW, ci.cout, 1, ci.k, ci.k), dtype=torch.uint8, device=device)
                wmax = max(wmax, max(1024, self.dw_bwd_parts) * ci.k * ci.k * ci.cout)      # wpartial[rows <= nparts][k*k][C]
            else:
                ci.w_fwd = torch.empty(nbytes(L.PACK_FWD, ci.cout, 27, 1, 1), dtype=torch.uint8, device=device)
                wmax = max(wmax, _STEM_WGRAD_PARTS_MAX * ci.cout * 27)   # k_stem_wgrad writes partial[nparts][Co][27]
            smax = max(smax, ci.cout)
        self.scratch_stats = torch.empty(_STATS_PARTS * 2 * smax, dtype=torch.float32, device=device)
        self.scratch_wgrad = torch.empty(wmax, dtype=torch.float32, device=device)
        self.scratch_wgrad2 = torch.empty(wmax, dtype=torch.float32, device=device)
        self.scratch_wgrad3 = torch.empty(wmax, dtype=torch.float32, device=device)      # main-stream producers rotate over 2..4:
        self.scratch_wgrad4 = torch.empty(wmax, dtype=torch.float32, device=device)      # a table is reduced up to two launches later
        self.scratch_red = torch.empty(_STATS_PARTS * 2 * smax, dtype=torch.float32, device=device)   # fused BN-bwd partials
        if self._ext_grad is not None:
            self.flat_grad = self._ext_grad
        else:
            self.flat_grad = torch.zeros(self.grad_numel, dtype=torch.float32, device=device)
        self.grad_views = []
        for ci in sorted(self.param_owners, key=lambda c: -c.stage):
            for j, p in enumerate(ci.params):
                o, n = ci.gslice[j]
                self.grad_views.append(self.flat_grad[o:o + n].view(p.shape))

    def _validate_modules(self):
        """The kernels read every parameter / buffer as dense fp32: anything else (model.half(), .bfloat16(),
        .double(), a non-contiguous view) must fail loudly instead of producing garbage."""
        for ci in self.convs:
            conv, bn = ci.mod.conv, ci.mod.bn
            named = [("conv.weight", conv.weight), ("conv.bias", conv.bias), ("bn.weight", bn.weight), ("bn.bias", bn.bias),
                     ("bn.running_mean", bn.running_mean), ("bn.running_var", bn.running_var)]
            for name, t in named:
                if t is None:
                    raise NotImplementedError("ConvBlock without %s" % name)
                if t.dtype != torch.float32 or not t.is_contiguous():
                    raise TypeError("mnasnet_pytorch_amd: %s must be contiguous float32 (got %s%s); the HIP path keeps fp32 "
                                    "master weights / statistics and bf16 activations internally -- do not call .half() / "
                                    ".bfloat16() / .double() on the model" %
                                    (name, t.dtype, "" if t.is_contiguous() else ", non-contiguous"))
            if bn.num_batches_tracked is not None and bn.num_batches_tracked.dtype != torch.int64:
                raise TypeError("bn.num_batches_tracked must be int64")
        for se in self.ses:
            for t in se.params:
                if t.dtype != torch.float32 or not t.is_contiguous():
                    raise TypeError("mnasnet_pytorch_amd: SqueezeExcite parameters must be contiguous float32")

    def _check_modes(self):
        """One mode per call: the launch list is compiled for root.training.  A submodule in a different mode (frozen-BN
        fine-tuning: ``bn.eval()`` under a training root) is not silently ignored."""
        mode = self.root.training
        for m in self.root.modules():
            if m.training != mode:
                raise NotImplementedError(
                    "mixed train/eval modes inside one engine subtree (%s.training=%s, root.training=%s): per-submodule "
                    "BatchNorm freezing is not supported by the HIP engine" % (type(m).__name__, m.training, mode))

    def bind_grad_buffer(self, buf: Optional[torch.Tensor]):
        """Make the kernels write gradients into ``buf`` (fp32, ``grad_numel`` elements, engine layout: later
        stages first) instead of an engine-owned buffer -- used by train_step.Trainer so that ONE flat buffer
        feeds the RCCL all-reduce buckets and the fused Adam."""
        if buf is not None and (buf.numel() != self.grad_numel or buf.dtype != torch.float32 or not buf.is_contiguous()):
            raise ValueError("grad buffer must be contiguous fp32 with %d elements" % self.grad_numel)
        self._ext_grad = buf
        self._sig = None           # forces _setup (programs hold gradient pointers)

    def ensure_setup(self, device):
        sig = self._signature()
        if self.lib is None or self.device != device or sig != self._sig:
            self._setup(device)
            self._sig = sig

    def reset_programs(self):
        """Drop the compiled launch lists (they embed the profiling brackets); rebuilt lazily on the next call.
        The HIP events they own are destroyed (after a device sync: a launch list may still be in flight)."""
        if any(p.busy for lst in self.programs.values() for p in lst):
            raise RuntimeError("reset_programs() while a forward is waiting for its backward")
        for lst in self.programs.values():
            for p in lst:
                p.destroy_graphs()            # syncs first when a program holds a graph executable
        self.programs.clear()
        self.profile_events = []
        if self._events:
            torch.cuda.synchronize(self.device)
            for h in self._events:
                self.lib.mnas_event_destroy(h)
            self._events = []

    def side_stream_handle(self):
        """hipStream_t for stream slot 1 of mnas_run_ops_multi.  The second stream exists only when use_side_stream asks for it
        (created on first use); otherwise slot 1 is the current stream too and launch lists carry no stream-1 ops."""
        if not self.use_side_stream:
            return L.cur_stream()
        if self.side_stream is None:
            self.side_stream = torch.cuda.Stream(device=self.device)
        return self.side_stream.cuda_stream

    def new_event(self):
        h = C.c_void_p()
        L.check(self.lib.mnas_event_create(C.byref(h)), "event_create")
        self._events.append(h.value)
        return h.value

    def __del__(self):
        try:
            for h in self._events:
                self.lib.mnas_event_destroy(h)
        except Exception:       # interpreter shutdown
            pass

    def read_profile(self):
        """[(tag, ms)] for every bracketed op launch since the programs were built (call after a sync)."""
        out = []
        for tag, e0, e1 in self.profile_events:
            ms = C.c_float()
            if self.lib.mnas_event_elapsed_ms(e0, e1, C.byref(ms)) == 0:
                out.append((tag, ms.value))
        return out

    def gptr(self, ci: _ConvInfo, j: int):
        return self.flat_grad.data_ptr() + 4 * ci.gslice[j][0]

    def set_input_normalization(self, mean=None, std=None):
        """Fuse the dataset's ``transforms.Normalize(mean, std)`` (datasets.py:474-516 with the constants of classifiers.py:91-92)
        into the stem conv's input load: float images are read as (x - mean) / std, uint8 images as (x / 255 - mean) / std
        (a quarter of the PCIe / HBM bytes of the fp32 batch train.py:427 uploads).  None, None removes the transform."""
        if not self.starts_with_stem:
            raise RuntimeError("input normalisation is fused into the stem conv: this engine does not start with one")
        if mean is None or std is None:
            self._in_norm = None
        else:
            m = torch.as_tensor(mean, dtype=torch.float64).flatten()
            s = torch.as_tensor(std, dtype=torch.float64).flatten()
            if m.numel() != 3 or s.numel() != 3 or bool((s <= 0).any()):
                raise ValueError("mean / std must have 3 entries, std > 0")
            self._in_norm = (m, s)
        self._in_aff = {}
        if self.programs:
            self.reset_programs()

    def input_affine(self, u8: bool):
        """device float[2][3] (scale, shift) of the stem's fused input transform for float / uint8 images, or None"""
        if getattr(self, "_in_norm", None) is None:
            return None
        t = self._in_aff.get((u8, self.device))
        if t is None:
            m, s = self._in_norm
            k = 255.0 if u8 else 1.0
            t = torch.stack([1.0 / (k * s), -m / s]).to(torch.float32).to(self.device).contiguous()
            self._in_aff[(u8, self.device)] = t
        return t

    def program(self, N, H, W, training, need_dx, pooled=False, in_u8=False) -> Program:
        key = (N, H, W, training, need_dx, pooled, bool(in_u8))
        lst = self.programs.setdefault(key, [])
        for p in lst:
            if not p.busy:
                return p
        if len(lst) >= _MAX_PROGRAMS_PER_SHAPE:
            raise RuntimeError(
                "%d forwards of shape %s are alive at once (their autograd graphs are still referenced and no backward "
                "has run): each holds a full set of activation buffers.  Drop the old outputs / call backward, or run "
                "under torch.no_grad()." % (len(lst), (N, self.in_channels_hint, H, W)))
        p = Program(self, N, H, W, training, need_dx, pooled, in_u8)
        lst.append(p)
        return p

    # ---- .grad bookkeeping (AccumulateGrad semantics on a flat buffer) -----------------------------
    def prepare_grads(self):
        """Returns True if every live .grad is already one of our views (accumulate in place)."""
        ours, other = 0, 0
        for p, v in zip(self.params, self.grad_views):
            if not p.requires_grad:
                continue
            if p.grad is not None and p.grad.data_ptr() == v.data_ptr() and p.grad.shape == v.shape:
                ours += 1
            else:
                other += 1
        if ours and other:       # mixed: zero the slices that are not live accumulators, then accumulate
            for p, v in zip(self.params, self.grad_views):
                if not (p.grad is not None and p.grad.data_ptr() == v.data_ptr()):
                    v.zero_()
            return True
        if ours:
            return True
        self.flat_grad.zero_()
        return False

    def finish_grads(self, accumulate):
        for p, v in zip(self.params, self.grad_views):
            if not p.requires_grad:
                continue
            if p.grad is None:
                p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():
                p.grad.add_(v)          # a foreign .grad tensor: add into it, like AccumulateGrad

    # ---- entry point ------------------------------------------------------------------------------
    def check_input(self, x: torch.Tensor):
        """Everything a launch list assumes about its input pointer; shared by forward() and Trainer.step's autograd-free path
        (a host or wrong-device pointer handed to the kernels is a GPU memory fault, not an exception)."""
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise RuntimeError("mnasnet_pytorch_amd runs on MI355X only: got a %s tensor; there is no CPU/eager "
                               "fallback (use oracle/ in tests for a CPU reference)" % (x.device if isinstance(x, torch.Tensor) else type(x)))
        if x.dim() != 4:
            raise ValueError("expected NCHW input")
        if x.shape[1] != (3 if self.starts_with_stem else self.in_channels_hint):
            raise ValueError("expected %d input channels, got %d" % (3 if self.starts_with_stem else self.in_channels_hint, x.shape[1]))
        if x.shape[0] < 1 or x.shape[2] < 1 or x.shape[3] < 1:
            raise ValueError("empty input %s" % (tuple(x.shape),))
        if any(p.device != x.device for p in self.params):
            raise RuntimeError("module parameters and input are on different devices")

    def forward(self, x: torch.Tensor, pooled: bool = False) -> torch.Tensor:
        """pooled=True returns the global average of the features, [N, C] fp32 (AdaptiveAvgPool2d(1) + flatten fused in)."""
        self.check_input(x)
        track = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.params))
        if x.dtype == torch.uint8 and self.starts_with_stem and getattr(self, "_in_norm", None) is not None:
            x = x.contiguous()              # uint8 images stay uint8: the stem converts and normalises on load
        else:
            x = x.float().contiguous()      # train.py:427 input.float()
        self.ensure_setup(x.device)
        self._check_modes()
        return _EngineFn.apply(self, track, bool(pooled), x, *self.params)
