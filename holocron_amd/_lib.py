"""ctypes binding of libholocron_hip.so (C ABI declared in include/holocron_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails, we raise.  The
product path is the HIP path; the CPU restatement under oracle/ is test infrastructure only.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HC_LIB_PATH") or os.path.join(_HERE, "lib", "libholocron_hip.so")      # HC_LIB_PATH: another build of the library (same-box A/Bs of kernel changes)

HC_MAX_TAPS = 12
HC_MT_CHUNK = 8192
HC_STAT_REPLICAS = 128

c_void_p, c_int32, c_int64, c_float, c_double = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double


class ConvClass(C.Structure):
    _fields_ = [("OHg", c_int32), ("OWg", c_int32), ("oy0", c_int32), ("ox0", c_int32), ("ostep", c_int32),
                ("istep", c_int32), ("ntaps", c_int32), ("tap", c_int32 * HC_MAX_TAPS)]


class ConvDesc(C.Structure):
    _fields_ = [("src0", c_void_p), ("src1", c_void_p), ("wpk", c_void_p), ("dst", c_void_p), ("resid", c_void_p),
                ("stats", c_void_p), ("bias", c_void_p), ("act", c_int32),
                ("N", c_int32), ("IH", c_int32), ("IW", c_int32), ("srcC", c_int32),
                ("OH", c_int32), ("OW", c_int32), ("Cout", c_int32), ("T", c_int32), ("nclass", c_int32),
                ("cls", ConvClass * 4), ("pix_scale", c_void_p), ("pix_shift", c_void_p), ("ch_coef", c_void_p),
                ("ch_mult", c_void_p), ("dst2", c_void_p), ("stats2", c_void_p), ("co_split", c_int32),
                ("ch_scale", c_void_p), ("act_slope", C.c_float), ("resid_after_act", c_int32)]


class ConvSmallDesc(C.Structure):
    _fields_ = [("srcA", c_void_p), ("srcB", c_void_p), ("w3", c_void_p), ("w1", c_void_p), ("out3", c_void_p),
                ("out1", c_void_p), ("resid", c_void_p), ("stats3", c_void_p), ("stats1", c_void_p),
                ("w3_rstride", c_int32), ("w1_rstride", c_int32),
                ("N", c_int32), ("H", c_int32), ("W", c_int32), ("C", c_int32), ("Cout", c_int32), ("mode", c_int32)]


class WgradDesc(C.Structure):
    _fields_ = [("x", c_void_p), ("dy", c_void_p), ("dw", c_void_p), ("ws", c_void_p),
                ("N", c_int32), ("IH", c_int32), ("IW", c_int32), ("Cin", c_int32), ("OH", c_int32), ("OW", c_int32),
                ("Cout", c_int32), ("KH", c_int32), ("KW", c_int32), ("stride", c_int32), ("pad", c_int32),
                ("beta", c_int32), ("co_valid", c_int32), ("ci_valid", c_int32)]


HC_WGRAD_MAX_JOBS = 16


class WgradGroupDesc(C.Structure):
    _fields_ = [("x", c_void_p * HC_WGRAD_MAX_JOBS), ("dy", c_void_p * HC_WGRAD_MAX_JOBS), ("dw", c_void_p * HC_WGRAD_MAX_JOBS),
                ("ws", c_void_p),
                ("njobs", c_int32), ("N", c_int32), ("IH", c_int32), ("IW", c_int32), ("Cin", c_int32), ("OH", c_int32), ("OW", c_int32),
                ("Cout", c_int32), ("KH", c_int32), ("KW", c_int32), ("stride", c_int32), ("pad", c_int32), ("beta", c_int32)]


class DwPackItem(C.Structure):
    _fields_ = [("w", c_void_p), ("out", c_void_p), ("C", c_int32), ("Cpad", c_int32), ("flip", c_int32), ("pad_", c_int32)]


HC_WREP_MAX_JOBS = 16


class RepWgradDesc(C.Structure):
    _fields_ = [("x", c_void_p * HC_WREP_MAX_JOBS), ("dy3", c_void_p * HC_WREP_MAX_JOBS), ("dy1", c_void_p * HC_WREP_MAX_JOBS),
                ("dw3", c_void_p * HC_WREP_MAX_JOBS), ("dw1", c_void_p * HC_WREP_MAX_JOBS), ("ws", c_void_p),
                ("njobs", c_int32), ("N", c_int32), ("IH", c_int32), ("IW", c_int32), ("Cin", c_int32), ("OH", c_int32),
                ("OW", c_int32), ("Cout", c_int32), ("stride", c_int32), ("accumulate", c_int32)]


class PackItem(C.Structure):
    _fields_ = [("w", c_void_p), ("dst", c_void_p), ("Cout", c_int32), ("Cin", c_int32), ("KH", c_int32),
                ("KW", c_int32), ("mode", c_int32), ("tap0", c_int32), ("T", c_int32), ("ld", c_int32)]


class ConvS2Desc(C.Structure):
    _fields_ = [("x", c_void_p), ("w3img", c_void_p), ("w1img", c_void_p), ("y3", c_void_p), ("y1", c_void_p),
                ("stats3", c_void_p), ("stats1", c_void_p), ("N", c_int32), ("H", c_int32), ("W", c_int32), ("Cin", c_int32),
                ("Cout", c_int32), ("x_nchw_f32", c_int32)]


class StemDesc(C.Structure):
    """hc_stem_desc (include/holocron_hip.h)."""
    _fields_ = [("x", c_void_p), ("w3img", c_void_p), ("w1img", c_void_p), ("N", c_int32), ("H", c_int32), ("W", c_int32)]


class StemBwdDesc(C.Structure):
    """hc_stem_bwd_desc (include/holocron_hip.h)."""
    _fields_ = [(n, c_void_p) for n in ("coef", "g", "save", "gamma3", "gamma1", "w3", "w1", "dgamma3", "dbeta3", "dgamma1", "dbeta1",
                                        "dw3", "dw1", "ws")] + [("act", c_int32), ("frozen", c_int32), ("accumulate", c_int32)]


class ConvS2DgradDesc(C.Structure):
    _fields_ = [("dy3", c_void_p), ("dy1", c_void_p), ("wimg", c_void_p), ("dx", c_void_p),
                ("N", c_int32), ("H", c_int32), ("W", c_int32), ("Cin", c_int32), ("Cout", c_int32)]


HC_MULTI_COPY_MAX = 64
HC_MULTI_COPY_PIECE = 262144


class MultiCopyDesc(C.Structure):
    _fields_ = [("src", c_void_p * HC_MULTI_COPY_MAX), ("dst", c_void_p * HC_MULTI_COPY_MAX), ("n", c_int64 * HC_MULTI_COPY_MAX),
                ("nitems", c_int32), ("src_bf16", c_int32), ("dst_bf16", c_int32), ("scale", C.c_float)]


class RepBnDesc(C.Structure):
    _fields_ = [("stats", c_void_p * 3), ("gamma", c_void_p * 3), ("beta", c_void_p * 3),
                ("running_mean", c_void_p * 3), ("running_var", c_void_p * 3), ("num_batches_tracked", c_void_p * 3),
                ("coef", c_void_p), ("save", c_void_p), ("C", c_int32), ("count", c_int64),
                ("eps", c_float), ("momentum", c_float), ("training", c_int32), ("c_valid", c_int32)]


class SeMlpDesc(C.Structure):
    """hc_se_mlp_desc (include/holocron_hip.h)."""
    _fields_ = [(n, c_void_p) for n in ("pooled", "w1", "gamma", "beta", "running_mean", "running_var", "num_batches_tracked", "w2",
                                        "b2", "h1", "part", "stat", "logits", "dl", "g", "part2", "dpool", "dw1", "dgamma", "dbeta",
                                        "dw2", "db2")] + \
               [("N", c_int32), ("C", c_int32), ("Cp", c_int32), ("R", c_int32), ("act", c_int32), ("eps", c_float),
                ("momentum", c_float)]


class RepBnBwdDesc(C.Structure):
    _fields_ = [("red", c_void_p), ("save", c_void_p), ("gamma", c_void_p * 3), ("dgamma", c_void_p * 3),
                ("dbeta", c_void_p * 3), ("bcoef", c_void_p), ("C", c_int32), ("count", c_int64),
                ("has_identity", c_int32), ("accumulate", c_int32), ("c_valid", c_int32), ("frozen", c_int32)]


class MtChunk(C.Structure):
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("m", c_void_p), ("s", c_void_p), ("smax", c_void_p),
                ("n", c_int32), ("group", c_int32), ("tensor", c_int32), ("flags", c_int32)]


class AdaBeliefGroup(C.Structure):
    _fields_ = [("lr", c_double), ("beta1", c_double), ("beta2", c_double), ("eps", c_double),
                ("weight_decay", c_double), ("step", c_int32), ("amsgrad", c_int32)]


class LarsGroup(C.Structure):
    _fields_ = [("lr", c_double), ("momentum", c_double), ("dampening", c_double), ("weight_decay", c_double),
                ("nesterov", c_int32), ("pad_", c_int32)]


class DropItem(C.Structure):
    _fields_ = [("off", c_int64), ("N", c_int32), ("H", c_int32), ("W", c_int32), ("block_size", c_int32), ("gamma", c_float),
                ("pad_", c_int32)]


class AdamxGroup(C.Structure):
    _fields_ = [("lr", c_double), ("beta1", c_double), ("beta2", c_double), ("beta3", c_double), ("alpha", c_double),
                ("eps", c_double), ("weight_decay", c_double), ("delta", c_double), ("step", c_int32), ("amsgrad", c_int32)]


HC_MSBN_MAX_BRANCHES = 6


class MsbnBranch(C.Structure):
    _fields_ = [("stats", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("running_mean", c_void_p), ("running_var", c_void_p),
                ("num_batches_tracked", c_void_p), ("dgamma", c_void_p), ("dbeta", c_void_p), ("stats_ld", c_int32),
                ("eps", c_float), ("momentum", c_float), ("pad_", c_int32)]


class MsbnDesc(C.Structure):
    _fields_ = [("br", MsbnBranch * HC_MSBN_MAX_BRANCHES), ("coef", c_void_p), ("save", c_void_p), ("red", c_void_p),
                ("bcoef", c_void_p), ("count", c_int64), ("B", c_int32), ("C", c_int32), ("c_valid", c_int32),
                ("training", c_int32), ("accumulate", c_int32), ("pad_", c_int32)]


class MsbnIo(C.Structure):
    _fields_ = [("y", c_void_p * HC_MSBN_MAX_BRANCHES), ("dy", c_void_p * HC_MSBN_MAX_BRANCHES),
                ("ld", c_int32 * HC_MSBN_MAX_BRANCHES), ("dld", c_int32 * HC_MSBN_MAX_BRANCHES), ("npix", c_int64),
                ("B", c_int32), ("C", c_int32)]


class LambGroup(C.Structure):
    _fields_ = [("lr", c_double), ("beta1", c_double), ("beta2", c_double), ("eps", c_double), ("weight_decay", c_double),
                ("clip_lo", c_double), ("clip_hi", c_double), ("rect", c_double), ("step", c_int32), ("mode", c_int32)]


def tap(dy, dx, src, wt):
    """HC_TAP of the header."""
    u = (dy & 0xff) | ((dx & 0xff) << 8) | ((src & 0xff) << 16) | ((wt & 0xff) << 24)
    return u - (1 << 32) if u >= (1 << 31) else u


# name -> (restype, argtypes); every symbol include/holocron_hip.h declares
SIGNATURES = {
    "hc_mixup": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_float, c_void_p]),
    "hc_mixup_onehot": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p]),
    "hc_topk_hits": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "hc_maxpool2_fwd": (c_int32, [c_void_p] * 3 + [c_int32] * 4 + [c_void_p]),
    "hc_maxpool2_bwd": (c_int32, [c_void_p] * 3 + [c_int32] * 4 + [c_void_p]),
    "hc_space_to_depth": (c_int32, [c_void_p] * 2 + [c_int32] * 8 + [c_void_p]),
    "hc_leaky_bwd": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p]),
    "hc_yolo1_loss_fwd": (c_int32, [c_void_p] * 3 + [c_int32] * 8 + [c_void_p] * 4 + [c_int32] + [c_void_p] * 4),
    "hc_yolo1_loss_bwd": (c_int32, [c_void_p] * 3 + [c_int32] * 8 + [c_void_p] * 4 + [c_int32] + [c_void_p] * 7),
    "hc_yolo1_decode": (c_int32, [c_void_p] * 3 + [c_int32] * 7 + [c_void_p] * 4),
    "hc_yolo_format_fwd": (c_int32, [c_void_p] * 2 + [c_int32] * 7 + [c_void_p] * 5),
    "hc_yolo_format_bwd": (c_int32, [c_void_p] * 3 + [c_int32] * 7 + [c_void_p] * 7),
    "hc_lamb_step": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "hc_tadam_step": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "hc_adan_step": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "hc_lookahead_sync": (c_int32, [c_void_p, c_int32, c_float, c_void_p]),
    "hc_multi_copy": (c_int32, [C.POINTER(MultiCopyDesc), c_void_p]),
    "hc_msbn_finalize": (c_int32, [C.POINTER(MsbnDesc), c_void_p]),
    "hc_msbn_bwd_finalize": (c_int32, [C.POINTER(MsbnDesc), c_void_p]),
    "hc_msbn_apply": (c_int32, [C.POINTER(MsbnIo), c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "hc_msbn_bwd_reduce": (c_int32, [C.POINTER(MsbnIo), c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p]),
    "hc_msbn_bwd_apply": (c_int32, [C.POINTER(MsbnIo), c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p]),
    "hc_dwrep_dgrad": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p] + [c_int32] * 5 + [c_void_p]),
    "hc_conv_gather": (c_int32, [C.POINTER(ConvDesc), c_void_p]),
    "hc_conv_pointwise_supported": (c_int32, [C.POINTER(ConvDesc)]),
    "hc_conv_pointwise": (c_int32, [C.POINTER(ConvDesc), c_void_p]),
    "hc_conv_small": (c_int32, [C.POINTER(ConvSmallDesc), c_void_p]),
    "hc_conv_small_supported": (c_int32, [C.POINTER(ConvSmallDesc)]),
    "hc_conv_s2_supported": (c_int32, [C.POINTER(ConvS2Desc)]),
    "hc_conv_s2_fwd": (c_int32, [C.POINTER(ConvS2Desc), c_void_p]),
    "hc_conv_s2_dgrad_supported": (c_int32, [C.POINTER(ConvS2DgradDesc)]),
    "hc_conv_s2_dgrad": (c_int32, [C.POINTER(ConvS2DgradDesc), c_void_p]),
    "hc_conv_s2_stem_wgrad_ws_bytes": (c_int64, []),
    "hc_conv_s2_stem_wgrad": (c_int32, [c_void_p] * 6 + [c_int32] * 4 + [c_void_p]),
    "hc_conv_wgrad_ws_bytes": (c_int64, [C.POINTER(WgradDesc)]),
    "hc_conv_wgrad": (c_int32, [C.POINTER(WgradDesc), c_void_p]),
    "hc_conv_wgrad_group_supported": (c_int32, [C.POINTER(WgradGroupDesc)]),
    "hc_conv_wgrad_group_ws_bytes": (c_int64, [C.POINTER(WgradGroupDesc)]),
    "hc_conv_wgrad_group": (c_int32, [C.POINTER(WgradGroupDesc), c_void_p]),
    "hc_rep_wgrad_supported": (c_int32, [C.POINTER(RepWgradDesc)]),
    "hc_rep_wgrad_ws_bytes": (c_int64, [C.POINTER(RepWgradDesc)]),
    "hc_rep_wgrad_plan": (c_int32, [C.POINTER(RepWgradDesc), c_void_p]),
    "hc_rep_wgrad": (c_int32, [C.POINTER(RepWgradDesc), c_void_p]),
    "hc_pack_conv_weight": (c_int32, [c_void_p, c_void_p] + [c_int32] * 7 + [c_void_p]),
    "hc_pack_conv_weights_multi": (c_int32, [c_void_p, c_int32, c_int64, c_void_p]),
    "hc_nchw_to_nhwc_bf16": (c_int32, [c_void_p, c_void_p] + [c_int32] * 5 + [c_void_p]),
    "hc_nhwc_bf16_to_nchw": (c_int32, [c_void_p, c_void_p] + [c_int32] * 5 + [c_void_p]),
    "hc_im2col_small": (c_int32, [c_void_p, c_void_p] + [c_int32] * 11 + [c_void_p]),
    "hc_unpack_im2col_grad": (c_int32, [c_void_p, c_void_p] + [c_int32] * 6 + [c_void_p]),
    "hc_rep_bn_finalize": (c_int32, [C.POINTER(RepBnDesc), c_void_p]),
    "hc_rep_apply": (c_int32, [c_void_p] * 6 + [c_int64, c_int32, c_int32, c_void_p]),
    "hc_channel_stats": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "hc_rep_bwd_reduce": (c_int32, [c_void_p] * 6 + [c_int64, c_int32, c_void_p]),
    "hc_rep_bwd_reduce_z": (c_int32, [c_void_p, c_void_p, c_int32] + [c_void_p] * 4 + [c_int64, c_int32, c_void_p]),
    "hc_rep_bwd_apply_z": (c_int32, [c_void_p, c_void_p, c_int32] + [c_void_p] * 7 + [c_int64, c_int32, c_void_p]),
    "hc_rep_bn_bwd_finalize": (c_int32, [C.POINTER(RepBnBwdDesc), c_void_p]),
    "hc_rep_bwd_apply": (c_int32, [c_void_p] * 9 + [c_int64, c_int32, c_void_p]),
    "hc_bn_act_apply": (c_int32, [c_void_p] * 3 + [c_int32] + [c_void_p] * 3 + [c_int32, c_int64, c_int32, c_int32, c_float, c_void_p]),
    "hc_bn_act_bwd_reduce": (c_int32, [c_void_p, c_int32] + [c_void_p] * 5 + [c_int64, c_int32, c_int32, c_float, c_void_p]),
    "hc_bn_act_bwd_apply": (c_int32, [c_void_p, c_int32] + [c_void_p] * 6 + [c_int64, c_int32, c_int32, c_float, c_void_p]),
    "hc_bn_act_apply_post": (c_int32, [c_void_p] * 3 + [c_int32] + [c_void_p] * 5 + [c_int32, c_int64, c_int32, c_int32, c_float, c_void_p]),
    "hc_bn_act_bwd_reduce_post": (c_int32, [c_void_p, c_int32] + [c_void_p] * 7 + [c_int64, c_int32, c_int32, c_float, c_void_p]),
    "hc_bn_act_bwd_apply_post": (c_int32, [c_void_p, c_int32] + [c_void_p] * 9 + [c_int64, c_int32, c_int32, c_float, c_void_p]),
    "hc_nhwc_copy": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32, c_int64, c_int32, c_void_p]),
    "hc_upsample2x_fwd": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32] + [c_int32] * 4 + [c_void_p]),
    "hc_upsample2x_bwd": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32] + [c_int32] * 4 + [c_void_p]),
    "hc_spp_fwd": (c_int32, [c_void_p] * 3 + [c_int32] * 4 + [c_void_p]),
    "hc_spp_bwd": (c_int32, [c_void_p] * 3 + [c_int32] * 4 + [c_void_p]),
    "hc_gap_fwd": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "hc_gap_bwd": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "hc_adabelief_step": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_void_p]),
    "hc_ademamix_step": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p]),
    "hc_adamp_step": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "hc_lars_step": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p]),
    "hc_hard_mish_fwd": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
    "hc_hard_mish_bwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "hc_box_pairwise": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "hc_box_pairwise_bwd": (c_int32, [c_void_p] * 5 + [c_int32, c_int32, c_int32, c_void_p]),
    "hc_nms_ws_bytes": (c_int64, [c_int32]),
    "hc_nms_sorted": (c_int32, [c_void_p, c_int32, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "hc_nms_sorted_batched": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "hc_focal_loss_fwd": (c_int32, [c_void_p] * 5 + [c_int32, c_int32, c_int64, c_int32, c_float, c_void_p]),
    "hc_focal_loss_bwd": (c_int32, [c_void_p] * 5 + [c_int32, c_int32, c_int64, c_float, c_void_p]),
    "hc_ce_fwd_bwd": (c_int32, [c_void_p] * 4 + [c_int32, c_int32, c_float, c_void_p]),
    "hc_ce_mean_fwd": (c_int32, [c_void_p] * 4 + [c_int32, c_int32, c_float, c_int64, c_void_p]),
    "hc_ce_mean_bwd": (c_int32, [c_void_p] * 5 + [c_int32, c_int32, c_float, c_int64, c_void_p]),
    "hc_ce_mean_aux_floats": (c_int64, [c_int32]),
    "hc_se_mlp_part_floats": (c_int64, [c_int32, c_int32]),
    "hc_se_mlp_fwd": (c_int32, [c_void_p, c_void_p]),
    "hc_se_mlp_bwd": (c_int32, [c_void_p, c_void_p]),
    "hc_poly_loss_hard_fwd": (c_int32, [c_void_p] * 5 + [c_int32, c_int32, c_int64, c_int32, c_float, c_void_p]),
    "hc_poly_loss_hard_bwd": (c_int32, [c_void_p] * 5 + [c_int32, c_int32, c_int64, c_float, c_void_p]),
    "hc_poly_loss_soft_fwd": (c_int32, [c_void_p] * 4 + [c_int32, c_int32, c_int64, c_int32, c_float, c_void_p]),
    "hc_poly_loss_soft_bwd": (c_int32, [c_void_p] * 5 + [c_int32, c_int32, c_int64, c_int32, c_float, c_void_p]),
    "hc_dice_sums": (c_int32, [c_void_p] * 3 + [c_int32, c_int32, c_int64, c_void_p]),
    "hc_dice_bwd": (c_int32, [c_void_p] * 3 + [c_int32, c_int32, c_int64, c_void_p]),
    "hc_dropblock_mask": (c_int32, [c_void_p] * 3 + [c_int32] * 4 + [c_float, c_void_p]),
    "hc_dropblock_mask_batched": (c_int32, [c_void_p, c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "hc_dropblock_apply": (c_int32, [c_void_p] * 4 + [c_int64, c_int32, c_int64, c_int32, c_int32, c_void_p]),
    "hc_yolo_decode": (c_int32, [c_void_p, c_int32, c_int64, c_int64, c_int64] + [c_int32] * 5 + [c_void_p, c_float]
                       + [c_void_p] * 4 + [c_int32, c_void_p]),
    "hc_yolo_assign": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p] + [c_int32] * 4 + [c_void_p] * 3),
    "hc_yolo_loss_fwd": (c_int32, [c_void_p, c_int32, c_int64, c_int64, c_int64] + [c_int32] * 5 + [c_void_p, c_float]
                         + [c_void_p] * 7),
    "hc_yolo_loss_bwd": (c_int32, [c_void_p, c_int32, c_int64, c_int64, c_int64] + [c_int32] * 5 + [c_void_p, c_float]
                         + [c_void_p] * 8),
    "hc_dw3x3_pack": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "hc_dw3x3_pack_multi": (c_int32, [c_void_p, c_int32, c_int32, c_void_p]),
    "hc_dw3x3_fwd": (c_int32, [c_void_p] * 4 + [c_int32] * 5 + [c_void_p]),
    "hc_dw3x3_dgrad": (c_int32, [c_void_p] * 4 + [c_int32] * 5 + [c_void_p]),
    "hc_dw3x3_wgrad_ws_bytes": (c_int64, [c_int32]),
    "hc_dw3x3_wgrad": (c_int32, [c_void_p] * 4 + [c_int32] * 7 + [c_void_p]),
    "hc_max_fwd": (c_int32, [c_void_p] * 3 + [c_int64, c_void_p]),
    "hc_max_bwd": (c_int32, [c_void_p] * 5 + [c_int64, c_void_p]),
    "hc_se_scale_fwd": (c_int32, [c_void_p] * 3 + [c_int64, c_int64, c_int32, c_int32, c_void_p]),
    "hc_se_scale_bwd_gate": (c_int32, [c_void_p] * 5 + [c_int64, c_int64, c_int32, c_int32, c_void_p]),
    "hc_se_scale_bwd_apply": (c_int32, [c_void_p] * 5 + [c_int64, c_int64, c_int32, c_int32, c_void_p]),
    "hc_slim_fold_fwd": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int64, c_int64, c_int32, c_void_p]),
    "hc_slim_fold_bwd_gate": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_int64,
                                        c_int32, c_void_p]),
    "hc_slim_fold_bwd_apply": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int64,
                                         c_int64, c_int32, c_void_p]),
    "hc_patch_stats": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p] + [c_int32] * 8 + [c_float, c_void_p]),
    "hc_normconv_bwd_scale": (c_int32, [c_void_p] * 5 + [c_int64, c_int32, c_void_p]),
    "hc_im2col_small_fp8": (c_int32, [c_void_p, c_void_p] + [c_int32] * 11 + [c_float, c_void_p]),
    "hc_quantize_fp8": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_int64, c_int32, c_float, c_void_p]),
    "hc_gap_fp8": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p]),
    "hc_stem_fused_supported": (c_int32, [c_void_p]),
    "hc_stem_stats": (c_int32, [c_void_p] * 4),
    "hc_stem_apply": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "hc_stem_bwd_ws_bytes": (c_int64, []),
    "hc_stem_bwd": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "hc_set_deterministic": (c_int32, [c_int32]),
    "hc_get_deterministic": (c_int32, []),
    "hc_get_stat_replicas": (c_int32, []),
    "hc_version": (C.c_char_p, []),
}

_lib = None


def load():
    """Load the shared library (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -m holocron_amd.build` "
            "(there is no CPU fallback for the HIP path)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library diverge
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class HipError(RuntimeError):
    pass


def stat_replicas() -> int:
    """Replicas of every per-channel statistics accumulator right now (128, or 32768 in deterministic mode)."""
    return int(load().hc_get_stat_replicas())


_REPLICA_LISTENERS = []      # called after the replica count changed: arenas sized for the old count start over


def on_replicas_changed(fn) -> None:
    _REPLICA_LISTENERS.append(fn)


def set_deterministic(on: bool) -> None:
    """Bit-reproducible training steps (include/holocron_hip.h hc_set_deterministic): every workgroup gets its own slot of the
    statistics accumulators, the split reductions become single-writer.  Costs memory (32768 replicas) and a slower finalize.
    The per-step zero arenas are told (``on_replicas_changed``) and start over, so that LEAVING the mode also returns its 256x
    larger zero-fill instead of clearing gigabytes on every later step."""
    before = stat_replicas()
    check(load().hc_set_deterministic(1 if on else 0), "hc_set_deterministic")
    if stat_replicas() != before:
        for fn in _REPLICA_LISTENERS:
            fn()


_ERR = {1: "bad argument", 2: "kernel launch failure"}


def check(rc, what):
    if rc != 0:
        raise HipError(f"{what} failed: {_ERR.get(rc, rc)}")


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise HipError("holocron_amd kernels need CUDA(HIP) tensors; got a CPU tensor "
                           "(the CPU restatement lives under oracle/ and is test-only)")
