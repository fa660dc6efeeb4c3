"""ctypes binding of libmnas_hip.so (include/mnas.h).  No CPU fallback: if the library is missing the
import of the product path fails loudly; if a launcher returns non-zero a RuntimeError is raised."""
from __future__ import annotations

import ctypes as C
import os

import torch  # imported first so that libamdhip64.so.7 resolves to the copy torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 8          # include/mnas.h: mnas_version()
LIB_PATH = os.environ.get("MNAS_LIB_PATH") or os.path.join(_HERE, "csrc", "libmnas_hip.so")      # override: A/B builds (tools/)

c_void_p, c_int, c_float, c_double, c_int64 = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_int64


class MnasActIn(C.Structure):
    _fields_ = [("data", c_void_p), ("scale", c_void_p), ("shift", c_void_p)]


class MnasGradIn(C.Structure):
    _fields_ = [("g", c_void_p), ("y", c_void_p), ("coef", c_void_p)]


class MnasConvGemm(C.Structure):
    _fields_ = [("mode", C.c_int32), ("N", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32), ("Ci", C.c_int32),
                ("Ho", C.c_int32), ("Wo", C.c_int32), ("Co", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
                ("stride", C.c_int32), ("pad", C.c_int32), ("nparts", C.c_int32), ("reserved", C.c_int32),
                ("act", MnasActIn), ("grad", MnasGradIn), ("w", c_void_p), ("bias", c_void_p), ("resid", c_void_p),
                ("out", c_void_p), ("stats", c_void_p), ("red_y", c_void_p), ("red_bn", c_void_p), ("gate", c_void_p)]


class MnasConvWgrad(C.Structure):
    _fields_ = [("N", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32), ("Ci", C.c_int32), ("Ho", C.c_int32),
                ("Wo", C.c_int32), ("Co", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32),
                ("pad", C.c_int32), ("nsplit", C.c_int32), ("x", MnasActIn), ("dy", MnasGradIn), ("partial", c_void_p)]


class MnasDwFwd(C.Structure):
    _fields_ = [("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("k", C.c_int32),
                ("nparts", C.c_int32), ("in_", MnasActIn), ("w", c_void_p), ("bias", c_void_p), ("out", c_void_p),
                ("stats", c_void_p), ("stride", C.c_int32), ("reserved", C.c_int32)]


class MnasDwBwd(C.Structure):
    _fields_ = [("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("k", C.c_int32),
                ("nparts", C.c_int32), ("x", MnasActIn), ("dy", MnasGradIn), ("w", c_void_p), ("gin", c_void_p),
                ("wpartial", c_void_p), ("red_bn", c_void_p), ("red_partial", c_void_p), ("phase", C.c_int32),
                ("stride", C.c_int32), ("g_masked", C.c_int32), ("reserved", C.c_int32)]


class MnasPwBwd(C.Structure):
    _fields_ = [("M", C.c_int32), ("Ci", C.c_int32), ("Co", C.c_int32), ("nparts", C.c_int32), ("x", MnasActIn),
                ("dy", MnasGradIn), ("w", c_void_p), ("resid", c_void_p), ("gin", c_void_p), ("wpartial", c_void_p),
                ("red_partial", c_void_p), ("red_y", c_void_p), ("red_bn", c_void_p), ("w_fwd", c_void_p),
                ("b_fwd", c_void_p), ("gin_masked", C.c_int32), ("seg_px", C.c_int32)]


class MnasPostWgrad(C.Structure):
    _fields_ = [("partial", c_void_p), ("grad", c_void_p), ("nsplit", C.c_int32), ("Co", C.c_int32), ("Ci", C.c_int32),
                ("taps", C.c_int32), ("dw", C.c_int32), ("level", C.c_int32)]


class MnasBwdPost(C.Structure):
    _fields_ = [("bn_partial", c_void_p), ("bnbuf", c_void_p), ("dgamma", c_void_p), ("dbeta", c_void_p), ("count", c_double),
                ("bn_nparts", C.c_int32), ("bn_C", C.c_int32), ("w1", MnasPostWgrad), ("w2", MnasPostWgrad)]


class MnasTconvDgrad(C.Structure):
    _fields_ = [("N", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32), ("Co", C.c_int32), ("Ci", C.c_int32),
                ("nparts", C.c_int32), ("dy", c_void_p), ("w", c_void_p), ("out", c_void_p), ("stats", c_void_p),
                ("red_y", c_void_p), ("red_bn", c_void_p)]


class MnasPackDesc(C.Structure):
    _fields_ = [("w", c_void_p), ("dst", c_void_p), ("kind", C.c_int32), ("Co", C.c_int32), ("Ci", C.c_int32),
                ("taps", C.c_int32)]


class MnasStemFwd(C.Structure):
    _fields_ = [("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
                ("Co", C.c_int32), ("nparts", C.c_int32), ("x", c_void_p), ("w", c_void_p), ("bias", c_void_p),
                ("out", c_void_p), ("stats", c_void_p), ("in_affine", c_void_p), ("in_u8", C.c_int32), ("reserved", C.c_int32)]


class MnasStemWgrad(C.Structure):
    _fields_ = [("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
                ("Co", C.c_int32), ("nparts", C.c_int32), ("x", c_void_p), ("dy", MnasGradIn), ("partial", c_void_p),
                ("in_affine", c_void_p), ("in_u8", C.c_int32), ("reserved", C.c_int32)]


class MnasOp(C.Structure):
    _fields_ = [("opcode", C.c_int32), ("i", C.c_int32 * 15), ("d", C.c_double * 4), ("p", c_void_p * 16)]


OP_CONV_GEMM, OP_CONV_WGRAD, OP_WGRAD_FINALIZE, OP_DW_FWD, OP_DW_BWD, OP_DW_WGRAD_FINALIZE = 1, 2, 3, 4, 5, 6
OP_STEM_FWD, OP_STEM_WGRAD, OP_BN_FWD_FINALIZE, OP_BN_BWD_REDUCE, OP_BN_BWD_FINALIZE = 7, 8, 9, 10, 11
OP_ADD_ACT, OP_NCHW_TO_NHWC, OP_PACK_WEIGHTS, OP_EVENT_RECORD, OP_EVENT_WAIT, OP_PW_BWD, OP_PACK_BATCH = 12, 13, 14, 15, 16, 17, 18
OP_POOL_ACT, OP_POOL_BWD, OP_DY_MAT = 22, 23, 24          # 19-21, 27-29, 36: retired in ABI 6 (include/mnas.h)
OP_BWD_POST, OP_TCONV_DGRAD = 25, 26
OP_HEAD_LINEAR, OP_SE_SCALE, OP_SE_BWD_REDUCE, OP_SE_BWD_APPLY = 30, 31, 32, 33
OP_SE_GATE, OP_SE_PROJ_FIN = 34, 35
OP_SE_FC_FWD, OP_SE_FC_BWD = 37, 38
OP_STEM_DGRAD = 39
PACK_FWD, PACK_DGRAD, PACK_DW, PACK_TCONV = 0, 1, 2, 3
EINVAL = 10001      # MNAS_EINVAL

# every symbol include/mnas.h declares: (name, restype, argtypes)
class MnasHeadLinear(C.Structure):
    _fields_ = [("N", C.c_int32), ("I", C.c_int32), ("O", C.c_int32), ("relu", C.c_int32), ("accumulate", C.c_int32),
                ("drop_p", c_float), ("seed", C.c_uint64), ("x", c_void_p), ("w", c_void_p), ("b", c_void_p),
                ("y", c_void_p), ("dz", c_void_p), ("dw", c_void_p), ("db", c_void_p), ("dx", c_void_p),
                ("relu_mask", c_void_p)]


SYMBOLS = {
    "mnas_conv_img_parts": (c_int, [c_int] * 11),
    "mnas_stem_parts": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "mnas_head_linear_fwd": (c_int, [C.POINTER(MnasHeadLinear), c_void_p]),
    "mnas_head_linear_bwd_w": (c_int, [C.POINTER(MnasHeadLinear), c_void_p]),
    "mnas_head_linear_bwd_x": (c_int, [C.POINTER(MnasHeadLinear), c_void_p]),
    "mnas_head_dropout_mask": (c_int, [c_void_p, c_int64, c_float, C.c_uint64, c_void_p]),
    "mnas_head_cross_entropy": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p]),
    "mnas_se_scale": (c_int, [C.POINTER(MnasActIn), c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mnas_se_bwd_reduce": (c_int, [c_void_p, C.POINTER(MnasActIn), c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "mnas_se_scratch_bytes": (c_int64, [c_int, c_int, c_int]),
    "mnas_se_bwd_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mnas_se_bwd_apply_cols": (c_int, [c_int, c_int, c_int]),
    "mnas_se_gate": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "mnas_se_proj_finalize": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "mnas_stem_dgrad": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p] * 3),
    "mnas_se_fc_supported": (c_int, [c_int, c_int]),
    "mnas_se_fc_fwd": (c_int, [c_void_p] * 5 + [c_int] * 3 + [c_void_p] * 4),
    "mnas_se_fc_bwd": (c_int, [c_void_p] * 5 + [c_int] * 3 + [c_void_p] * 6 + [c_int, c_void_p]),
    "mnas_conv_gemm_gate_ok": (c_int, [c_int, c_int, c_int, c_int]),
    "mnas_graph_create": (c_int, [C.POINTER(MnasOp), c_int, C.POINTER(c_void_p), c_int, C.POINTER(c_void_p), C.POINTER(c_int)]),
    "mnas_graph_launch": (c_int, [c_void_p, c_void_p]),
    "mnas_graph_destroy": (c_int, [c_void_p]),
    "mnas_version": (c_int, []),
    "mnas_arch": (C.c_char_p, []),
    "mnas_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int]),
    "mnas_conv_gemm": (c_int, [C.POINTER(MnasConvGemm), c_void_p]),
    "mnas_conv_gemm_tile_pixels": (c_int, [c_int, c_int, c_int]),
    "mnas_conv_gemm_parts": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "mnas_conv_wgrad": (c_int, [C.POINTER(MnasConvWgrad), c_void_p]),
    "mnas_conv_wgrad_slabs": (c_int, [c_int, c_int, c_int]),
    "mnas_wgrad_finalize": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "mnas_pw_bwd": (c_int, [C.POINTER(MnasPwBwd), c_void_p]),
    "mnas_pw_bwd_supported": (c_int, [c_int, c_int]),
    "mnas_pw_bwd_forms": (c_int, [c_int, c_int]),
    "mnas_pw_bwd_tile_pixels": (c_int, [c_int, c_int]),
    "mnas_pw_bwd_slices": (c_int, [c_int, c_int]),
    "mnas_dw_fwd": (c_int, [C.POINTER(MnasDwFwd), c_void_p]),
    "mnas_dw_bwd": (c_int, [C.POINTER(MnasDwBwd), c_void_p]),
    "mnas_dw_rows": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "mnas_dw_geometry": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, C.POINTER(c_int)]),
    "mnas_dw_wgrad_finalize": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "mnas_stem_fwd": (c_int, [C.POINTER(MnasStemFwd), c_void_p]),
    "mnas_stem_wgrad": (c_int, [C.POINTER(MnasStemWgrad), c_void_p]),
    "mnas_bn_fwd_finalize": (c_int, [c_void_p, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_float, c_float, c_int, c_void_p, c_void_p]),
    "mnas_bn_bwd_reduce": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "mnas_tconv_dgrad": (c_int, [C.POINTER(MnasTconvDgrad), c_void_p]),
    "mnas_tconv_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "mnas_tconv_parts": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "mnas_bwd_post": (c_int, [C.POINTER(MnasBwdPost), c_void_p]),
    "mnas_dy_materialize": (c_int, [C.POINTER(MnasGradIn), c_int64, c_int, c_void_p, c_void_p]),
    "mnas_bn_bwd_finalize": (c_int, [c_void_p, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "mnas_add_act": (c_int, [C.POINTER(MnasActIn), C.POINTER(MnasActIn), c_int64, c_int, c_void_p, c_void_p, c_int,
                             c_void_p]),
    "mnas_pool_act": (c_int, [C.POINTER(MnasActIn), c_int, c_int, c_int, c_void_p, c_void_p]),
    "mnas_pool_bwd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mnas_nchw_f32_to_nhwc_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mnas_pack_weights": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mnas_pack_weights_batch": (c_int, [c_void_p, c_int, c_void_p]),
    "mnas_packed_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "mnas_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float,
                               c_float, c_int, c_float, c_void_p]),
    "mnas_rmsprop_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float,
                                  c_float, c_void_p]),
    "mnas_sgd_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_int, c_int, c_float,
                              c_void_p]),
    "mnas_run_ops": (c_int, [C.POINTER(MnasOp), c_int, c_void_p, C.POINTER(c_int)]),
    "mnas_run_ops_multi": (c_int, [C.POINTER(MnasOp), c_int, C.POINTER(c_void_p), c_int, C.POINTER(c_int)]),
    "mnas_event_create": (c_int, [C.POINTER(c_void_p)]),
    "mnas_event_destroy": (c_int, [c_void_p]),
    "mnas_event_record": (c_int, [c_void_p, c_void_p]),
    "mnas_event_elapsed_ms": (c_int, [c_void_p, c_void_p, C.POINTER(c_float)]),
    "mnas_probe_copy": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "mnas_probe_valu": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "mnas_probe_copy4": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mnas_probe_read": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mnas_probe_empty": (c_int, [c_int, c_int, c_void_p]),
}

_lib = None


def load():
    """Load libmnas_hip.so and type every entry point.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libmnas_hip.so not built (%s): run `make -C mnasnet_pytorch_amd/csrc` or __graft_entry__.build(). "
            "There is no CPU / eager fallback for the MNASNet hot path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)     # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    if lib.mnas_version() != ABI_VERSION:
        raise RuntimeError("libmnas_hip.so ABI version %d != %d (stale build: run make -C mnasnet_pytorch_amd/csrc)" % (lib.mnas_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what="mnas call"):
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (what, rc))


def cur_stream():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()
