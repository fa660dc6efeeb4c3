"""Helpers for the -m gpu parity tests: thin torch-tensor wrappers over the C ABI (include/mnas.h)."""
import ctypes as C

import torch

from mnasnet_pytorch_amd import _lib as L


def bf16r(t):
    return t.to(torch.bfloat16).to(torch.float32)


def nhwc(x):
    """NCHW fp32 (cpu) -> NHWC bf16 cuda"""
    return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()


def from_nhwc(t):
    """NHWC bf16 cuda -> NCHW fp32 cpu"""
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def relerr(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def act_in(data, scale=None, shift=None):
    return L.MnasActIn(L.ptr(data), L.ptr(scale), L.ptr(shift))


def grad_in(g, y, coef):
    return L.MnasGradIn(L.ptr(g), L.ptr(y), L.ptr(coef))


def pack(w, kind):
    """w: reference-layout fp32 weight (cpu or cuda) -> packed device buffer"""
    lib = L.load()
    w = w.detach().float().cuda().contiguous()
    Co, Cig, kh, kw = w.shape
    Ci = Cig
    nbytes = lib.mnas_packed_bytes(kind, Co, Ci, kh, kw)
    dst = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    L.check(lib.mnas_pack_weights(w.data_ptr(), kind, Co, Ci, kh, kw, dst.data_ptr(), L.cur_stream()), "pack")
    return dst


def conv_gemm(mode, N, Hi, Wi, Ci, Ho, Wo, Co, k, stride, pad, w, bias=None, act=None, grad=None, resid=None,
              nparts=64, stats=False, red_y=None, red_bn=None, gate=None, expect=0):
    lib = L.load()
    out = torch.empty((N, Ho, Wo, Co), dtype=torch.bfloat16, device="cuda")
    st = torch.full((2, Co, nparts), float("nan"), device="cuda") if stats else None
    a = L.MnasConvGemm()
    a.mode, a.N, a.Hi, a.Wi, a.Ci, a.Ho, a.Wo, a.Co = mode, N, Hi, Wi, Ci, Ho, Wo, Co
    a.kh = a.kw = k
    a.stride, a.pad, a.nparts = stride, pad, nparts
    if act is not None:
        a.act = act
    if grad is not None:
        a.grad = grad
    a.w, a.bias, a.resid, a.out, a.stats = L.ptr(w), L.ptr(bias), L.ptr(resid), L.ptr(out), L.ptr(st)
    a.red_y, a.red_bn = L.ptr(red_y), L.ptr(red_bn)
    a.gate = L.ptr(gate)
    rc = lib.mnas_conv_gemm(C.byref(a), L.cur_stream())
    if expect:
        assert rc == expect, rc
        return None, None
    L.check(rc, "conv_gemm")
    return out, st


def rand_bn_coefs(C, seed, O):
    """Plausible (s,t,c1,c2,c3,mean,invstd,_) rows for dy-on-load tests; returns [8][C] fp32 cpu"""
    u = O.det_uniform((8, C), seed)
    b = torch.zeros(8, C)
    b[0] = 1.0 + 0.3 * u[0]          # s
    b[1] = 0.2 * u[1]                # t
    b[2] = b[0]                      # c1 = s
    b[3] = 0.05 * u[3]               # c2
    b[4] = 0.02 * u[4]               # c3
    b[5] = 0.1 * u[5]                # mean
    b[6] = 1.0 + 0.2 * u[6].abs()    # invstd
    return b


def dy_ref(g, y, b, rounded=True):
    """g,y: NCHW fp32 (already bf16-rounded); b: [8][C].  Returns dy as the kernels use it: bf16-rounded where it is
    staged into LDS for the MFMA kernels, fp32 for the depthwise kernels (which form it on the fly from raw g, y)."""
    s, t, c1, c2, c3 = (b[i].view(1, -1, 1, 1) for i in range(5))
    dz = g * ((s * y + t) > 0)
    d = c1 * dz + c2 * y + c3
    return bf16r(d) if rounded else d
