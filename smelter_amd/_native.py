"""ctypes binding of libsmelter_hip.so (include/smelter_hip.h).

There is NO fallback: if the library is missing or fails to load, importing this module raises — the product
path never routes through PyTorch ops or the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SMELTER_LIB_PATH: another BUILD of the same library for same-box A/B runs and diagnostic builds — tools/lib_ab.sh; never a fallback: a missing file raises)
LIB_PATH = os.environ.get("SMELTER_LIB_PATH") or os.path.join(_HERE, "libsmelter_hip.so")


class SmelterError(RuntimeError):
    """Mirror of ONNXGraph.Errors (/root/reference/Sources/Smelter/ONNXGraph.swift:38-47) + library codes."""
    KINDS = {-1: "unsupportedInput", -2: "unsupportedOutput", -3: "unknownNodeOpType", -4: "noSuchOutput",
             -5: "graphInternalError", -6: "insufficientInputs", -7: "inconsistentState",
             -8: "notEnoughAttributes", -9: "protoDecode", -10: "unsupportedDtype", -11: "flattenAxis",
             -12: "hip", -13: "invalidArgument"}

    def __init__(self, code: int, detail: str):
        self.code = code
        self.kind = self.KINDS.get(code, "unknown")
        self.detail = detail
        super().__init__("%s (%d): %s" % (self.kind, code, detail))


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("N", "H", "W", "Cin", "Cout", "kH", "kW", "sH", "sW", "dH", "dW",
                                         "pt", "pl", "pb", "pr", "groups", "Ho", "Wo", "act")] + \
               [("slope", C.c_float)] + [(n, C.c_int32) for n in ("x_ld", "y_ld", "res_ld", "y_sH", "y_sW", "y_oH", "y_oW", "y_H", "y_W", "x_ups")]


class StagePhase(C.Structure):
    """sm_stage_phase (include/smelter_hip.h): one convolution of a persistent run"""
    _fields_ = [("d", ConvDesc), ("x", C.c_void_p), ("w_ohwi", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p), ("y", C.c_void_p), ("w_packed", C.c_void_p),
                ("x2", C.c_void_p), ("K2", C.c_int32), ("x2_ld", C.c_int32), ("x2_H", C.c_int32), ("x2_W", C.c_int32), ("x2_stride", C.c_int32), ("form", C.c_int32)]


class MbConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("N", "H", "W", "Ho", "Wo", "stride", "Cin", "Ch", "Cout", "expand", "residual")] + \
               [("act1", C.c_int32), ("slope1", C.c_float), ("act2", C.c_int32), ("slope2", C.c_float), ("act3", C.c_int32), ("slope3", C.c_float),
                ("x_ld", C.c_int32), ("y_ld", C.c_int32)]


class PoolDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("N", "H", "W", "C", "kH", "kW", "sH", "sW", "y0", "x0", "Ho", "Wo",
                                         "x_ld", "y_ld")]


class Config(C.Structure):
    _fields_ = [("input_constraint", C.c_int32), ("align_corners", C.c_int32), ("dims", C.c_int32 * 4),
                ("fuse", C.c_int32), ("use_hip_graph", C.c_int32), ("autotune", C.c_int32), ("keep_intermediates", C.c_int32),
                ("branch_concurrency", C.c_int32), ("fusions", C.c_int32), ("autotune_mode", C.c_int32), ("autotune_max_cfg", C.c_int32),
                ("mbconv_domain", C.c_int32 * 3)]


class Shape(C.Structure):
    _fields_ = [("channels", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("depth", C.c_int32)]


# every symbol include/smelter_hip.h declares: name -> (restype, argtypes)
_vp, _i32, _i64, _f, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
_pi32 = C.POINTER(C.c_int32)
class BneckDesc(C.Structure):
    """sm_bneck_desc (include/smelter_hip.h): a bottleneck block launched as one tile pass (sm_bottleneck_fwd)"""
    _fields_ = [("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C0", C.c_int32), ("Cm", C.c_int32), ("C3", C.c_int32),
                ("x_ld", C.c_int32), ("y_ld", C.c_int32), ("res_ld", C.c_int32),
                ("act1", C.c_int32), ("slope1", C.c_float), ("act2", C.c_int32), ("slope2", C.c_float), ("act3", C.c_int32), ("slope3", C.c_float)]


class RunInfo(C.Structure):
    """smelter_run_info (include/smelter_hip.h): one persistent run of an executor and what the compile's in-situ A/B measured for it"""
    _fields_ = [("first_step", C.c_int32), ("layers", C.c_int32), ("on", C.c_int32), ("timed", C.c_int32), ("samples", C.c_int32), ("mode_bits", C.c_int32),
                ("err_bits", C.c_int32), ("why_off", C.c_int32), ("t_run_us", C.c_float), ("t_layers_us", C.c_float)]


SYMBOLS = {
    "smelter_last_error_string": (C.c_char_p, []),
    "smelter_version": (C.c_char_p, []),
    "sm_conv2d_fwd": (_i32, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp]),
    "sm_conv2d_workspace_bytes": (_sz, [C.POINTER(ConvDesc)]),
    "sm_conv2d_uses_tile_config": (_i32, [C.POINTER(ConvDesc)]),
    "sm_conv2d_fwd_ws": (_i32, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sm_conv2d_num_configs": (_i32, []),
    "sm_conv2d_config_name": (C.c_char_p, [_i32]),
    "sm_conv2d_config_loader_waves": (_i32, [_i32]),
    "sm_conv2d_fwd_cfg": (_i32, [C.POINTER(ConvDesc), _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sm_conv2d_dual_fwd_cfg": (_i32, [C.POINTER(ConvDesc), _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "sm_stage_supported": (_i32, [C.POINTER(StagePhase), _i32]),
    "sm_stage_device_fast": (_i32, []),
    "sm_stage_phase_kind": (_i32, [C.POINTER(StagePhase), _i32, _i32]),
    "sm_stage_ctl_bytes": (_sz, [C.POINTER(StagePhase), _i32]),
    "sm_stage_fwd": (_i32, [C.POINTER(StagePhase), _i32, _vp, _vp]),
    "sm_stage_packed_weight_bytes": (_sz, [C.POINTER(StagePhase), _i32, _i32]),
    "sm_stage_pack_weights": (_i32, [C.POINTER(StagePhase), _i32, _i32, _vp, _vp]),
    "sm_stage_status": (_i32, [_vp, _vp, _pi32, _pi32]),
    "sm_conv2d_pool_supported": (_i32, [C.POINTER(ConvDesc)]),
    "sm_conv2d_pool_fwd_cfg": (_i32, [C.POINTER(ConvDesc), _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "sm_conv2d_head_supported": (_i32, [C.POINTER(ConvDesc), _i32]),
    "sm_conv2d_head_fwd_cfg": (_i32, [C.POINTER(ConvDesc), _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f, _vp, _i32, _vp]),
    "sm_conv2d_stem_f32_supported": (_i32, [C.POINTER(ConvDesc), _i32, _i32, _i32]),
    "sm_conv2d_stem_f32_pool_supported": (_i32, [C.POINTER(ConvDesc), _i32, _i32, _i32, _i32, _i32]),
    "sm_conv2d_stem_f32_pool_fwd": (_i32, [C.POINTER(ConvDesc), _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "sm_conv2d_stem_f32_fwd_cfg": (_i32, [C.POINTER(ConvDesc), _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "sm_conv2d_xhead_supported": (_i32, [C.POINTER(ConvDesc), _i32, _i32]),
    "sm_bottleneck_supported": (_i32, [C.POINTER(BneckDesc)]),
    "sm_bottleneck_fwd": (_i32, [C.POINTER(BneckDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sm_conv2d_xhead_proj_supported": (_i32, [C.POINTER(ConvDesc), _i32]),
    "sm_conv2d_xhead_proj_fwd_cfg": (_i32, [C.POINTER(ConvDesc), _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f, _vp, _i32, _vp, _vp, _i32, _vp]),
    "sm_conv2d_xhead_fwd_cfg": (_i32, [C.POINTER(ConvDesc), _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f, _vp, _i32, _vp, _i32, _vp]),
    "sm_mbconv_supported": (_i32, [C.POINTER(MbConvDesc)]),
    "sm_mbconv_kernel_kind": (_i32, [C.POINTER(MbConvDesc)]),
    "sm_mbconv_lds_bytes": (_sz, [C.POINTER(MbConvDesc)]),
    "sm_mbconv_fwd": (_i32, [C.POINTER(MbConvDesc)] + [_vp] * 9),
    "sm_mbconv_fwd_kind": (_i32, [C.POINTER(MbConvDesc), _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sm_mbconv_stem_supported": (_i32, [C.POINTER(MbConvDesc)] + [_i32] * 5),
    "sm_mbconv_stem_fwd": (_i32, [C.POINTER(MbConvDesc), _vp] + [_i32] * 5 + [_vp] * 8),
    "sm_stem3x3_f32_supported": (_i32, [_i32] * 7),
    "sm_stem3x3_f32_fwd": (_i32, [_vp] + [_i32] * 4 + [_vp, _vp] + [_i32] * 4 + [_f, _vp, _i32, _vp]),
    "sm_conv_transpose2d_fwd": (_i32, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp]),
    "sm_conv2d_time": (_i32, [C.POINTER(ConvDesc), _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i32, _vp, C.POINTER(_f)]),
    "sm_batchnorm": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _i32, _f, _vp]),
    "sm_add": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _f, _vp]),
    "sm_binary": (_i32, [_i32, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _f, _vp]),
    "sm_neuron": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _f, _vp]),
    "sm_softmax": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "sm_group_norm": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _f, _i32, _i32, _vp]),
    "sm_pad2d": (_i32, [_vp, _vp] + [_i32] * 11 + [_vp]),
    "sm_relu": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _f, _vp]),
    "sm_maxpool2d": (_i32, [C.POINTER(PoolDesc), _vp, _vp, _vp]),
    "sm_avgpool2d": (_i32, [C.POINTER(PoolDesc), _vp, _vp, _vp]),
    "sm_global_avgpool": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "sm_upsample_nearest": (_i32, [_vp, _vp] + [_i32] * 8 + [_vp]),
    "sm_upsample_bilinear": (_i32, [_vp, _vp] + [_i32] * 9 + [_vp]),
    "sm_copy_channels": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "sm_reshape_nchw_order": (_i32, [_vp, _vp] + [_i32] * 7 + [_vp]),
    "sm_nchw_f32_to_nhwc_f16": (_i32, [_vp, _vp] + [_i32] * 9 + [_vp]),
    "sm_resize_nchw_f32": (_i32, [_vp, _vp] + [_i32] * 6 + [_vp]),
    "sm_nchw_f32_to_rowpair_f16": (_i32, [_vp, _vp] + [_i32] * 8 + [_vp]),
    "sm_nhwc_f16_to_nchw_f32": (_i32, [_vp, _vp] + [_i32] * 5 + [_vp]),
    "sm_nhwc_f16_to_float_array": (_i32, [_vp, _vp] + [_i32] * 5 + [_vp]),
    "sm_float_array_count": (_sz, [_i32] * 4),
    "sm_half_to_float": (None, [_vp, _vp, _sz]),
    "sm_float_to_half": (None, [_vp, _vp, _sz]),
    "sm_reformat_conv_weight": (_i32, [_vp, _vp] + [_i32] * 6),
    "sm_conv_padded_size": (None, [_i32] * 13 + [_pi32, _pi32]),
    "sm_pool_padded_size": (None, [_i32] * 8 + [_pi32, _pi32]),
    "sm_pool_window_start": (_i32, [_i32] * 4),
    "sm_device_count": (_i32, [_pi32]),
    "sm_set_device": (_i32, [_i32]),
    "sm_malloc": (_i32, [C.POINTER(_vp), _sz]),
    "sm_free": (_i32, [_vp]),
    "sm_memcpy_h2d": (_i32, [_vp, _vp, _sz, _vp]),
    "sm_memcpy_d2h": (_i32, [_vp, _vp, _sz, _vp]),
    "sm_memset": (_i32, [_vp, _i32, _sz, _vp]),
    "sm_stream_create": (_i32, [C.POINTER(_vp)]),
    "sm_stream_destroy": (_i32, [_vp]),
    "sm_stream_synchronize": (_i32, [_vp]),
    "sm_timer_create": (_i32, [C.POINTER(_vp)]),
    "sm_timer_start": (_i32, [_vp, _vp]),
    "sm_timer_stop": (_i32, [_vp, _vp]),
    "sm_timer_elapsed_ms": (_i32, [_vp, C.POINTER(_f)]),
    "sm_timer_destroy": (_i32, [_vp]),
    "smelter_config_default": (None, [C.POINTER(Config)]),
    "smelter_graph_create": (_i32, [_vp, _sz, C.POINTER(Config), C.POINTER(_vp)]),
    "smelter_graph_destroy": (None, [_vp]),
    "smelter_graph_format": (_i32, [_vp]),
    "smelter_graph_set_format": (_i32, [_vp, _i32]),
    "smelter_graph_output_shapes": (_i32, [_vp, C.POINTER(Shape), _i32]),
    "smelter_graph_num_nodes": (_i32, [_vp]),
    "smelter_graph_weight_bytes": (_i32, [_vp, _i32, C.POINTER(_sz)]),
    "smelter_graph_pack_weights": (_i32, [_vp, _i32, _vp, _sz]),
    "smelter_graph_compile": (_i32, [_vp, _i32, _i32, _vp, _i32, C.POINTER(_vp)]),
    "smelter_graph_compile_with_plan": (_i32, [_vp, _i32, _i32, _vp, _i32, C.POINTER(C.c_int32), C.c_size_t, C.POINTER(_vp)]),
    "smelter_exec_get_plan": (_i32, [_vp, C.POINTER(C.c_int32), C.c_size_t, C.POINTER(C.c_size_t)]),
    "smelter_graph_default_plan": (_i32, [_vp, _i32, C.POINTER(C.c_int32), C.c_size_t, C.POINTER(C.c_size_t)]),
    "smelter_exec_destroy": (None, [_vp]),
    "smelter_exec_encode_sized": (_i32, [_vp, _vp, _vp, _i32, _i32]),
    "smelter_exec_encode": (_i32, [_vp, _vp, _vp]),
    "smelter_exec_output_dims": (_i32, [_vp, _pi32, _pi32, _pi32, _pi32]),
    "smelter_exec_read_float": (_i32, [_vp, _vp, _vp, _sz, _i32]),
    "smelter_exec_output_device": (_vp, [_vp]),
    "smelter_exec_num_launches": (_i32, [_vp]),
    "smelter_exec_num_ops": (_i32, [_vp]),
    "smelter_exec_num_runs": (_i32, [_vp]),
    "smelter_exec_run_info": (_i32, [_vp, _i32, C.POINTER(RunInfo)]),
    "smelter_exec_weight_bytes": (_sz, [_vp]),
    "smelter_exec_weight_arena": (_vp, [_vp]),
    "smelter_exec_activation_bytes": (_sz, [_vp]),
    "smelter_exec_flops": (C.c_double, [_vp]),
    "smelter_exec_algorithmic_bytes": (C.c_double, [_vp]),
    "smelter_exec_launch_bytes": (C.c_double, [_vp]),
    "smelter_exec_op_desc": (C.c_char_p, [_vp, _i32]),
    "smelter_exec_num_values": (_i32, [_vp]),
    "smelter_exec_value_name": (C.c_char_p, [_vp, _i32]),
    "smelter_exec_value_stored": (_i32, [_vp, _i32]),
    "smelter_exec_time_op": (_i32, [_vp, _vp, _i32, _i32, C.POINTER(_f)]),
    "smelter_exec_profile": (_i32, [_vp, _vp, _vp, _i32, C.POINTER(_f), _i32]),
    "smelter_exec_read_value": (_i32, [_vp, _vp, C.c_char_p, _vp, _sz, _pi32, _pi32, _pi32]),
}

_lib = None


def lib():
    """Load libsmelter_hip.so.  torch is imported first so that one HIP runtime (the one torch already mapped,
    SONAME libamdhip64.so.7) serves both torch's allocator/streams and our kernels."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libsmelter_hip.so is not built (run `make` or __graft_entry__.build()); "
                              "there is no fallback path")
        try:
            import torch  # noqa: F401  (plumbing: device memory, streams, torch.distributed)
        except Exception:  # pragma: no cover - torch is optional for the pure C-ABI users
            pass
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError here == header/library drift: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int):
    if rc != 0:
        raise SmelterError(rc, lib().smelter_last_error_string().decode("utf-8", "replace"))
