"""ctypes binding of liblt_hip.so (include/lt_hip.h).  PyTorch appears here only as the owner of device
memory and streams: every function takes raw ``data_ptr()`` addresses and a ``hipStream_t``.

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os

# The HIP runtime maps every hipStream onto one of GPU_MAX_HW_QUEUES hardware queues (default 4); two streams that land on the same queue run one after the
# other.  Measured (round 5, tools/concurrency_probe.py): two forwards of two model instances on two streams took exactly the sum of their times with the
# default, and overlapped completely with 8 queues (B = 1 per forward: 8.2 -> 4.7 ms for both).  The training step (main + weight-gradient side stream +
# RCCL's own streams) and concurrent inference requests want distinct queues, so ask for 8 unless the caller has decided otherwise.  Only effective when
# this module is imported before the process's first HIP call (the variable is read when the runtime initialises).
# Opt-out: LT_KEEP_HW_QUEUES=1 leaves the variable alone (a host application that manages its own runtime configuration); an explicit GPU_MAX_HW_QUEUES in
# the environment always wins.  If HIP was initialised before this import the variable can no longer take effect: say so instead of failing silently
# (INTEGRATION.md "Process-wide settings").
import sys
import warnings


def _request_hw_queues():
    if os.environ.get("LT_KEEP_HW_QUEUES") == "1" or "GPU_MAX_HW_QUEUES" in os.environ:
        return
    t = sys.modules.get("torch")
    if t is not None and t.cuda.is_initialized():
        warnings.warn("lt_hip: HIP is already initialised, GPU_MAX_HW_QUEUES=8 cannot take effect any more (streams may share hardware queues: the "
                      "training step's side stream and concurrent forwards then serialise).  Import lt_hip before the first CUDA/HIP call or export "
                      "GPU_MAX_HW_QUEUES=8 yourself; LT_KEEP_HW_QUEUES=1 silences this.", RuntimeWarning, stacklevel=3)
        return
    os.environ["GPU_MAX_HW_QUEUES"] = "8"


_request_hw_queues()

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
# LT_HIP_LIB: load an A/B variant build of the same ABI instead (lt_build.build_variant); never a different backend
LIB_PATH = os.environ.get("LT_HIP_LIB") or os.path.join(HERE, "lib", "liblt_hip.so")

LT_F32, LT_BF16, LT_FP8 = 0, 1, 2
AGG = {"sum": 0, "max": 1, "softmax": 2, "conf": 3, "conf_norm": 4}
EPI_RELU_PRE, EPI_RELU_POST, EPI_STORE_F32, EPI_SIGMOID = 1, 2, 4, 8
EPI_RES_F32 = 64
BN_FROZEN = 32
BN_Y_BF16 = 128
ACT_BF16 = 256
TILE_AUTO, TILE_128x128, TILE_128x64, TILE_256x32, TILE_256x16, TILE_64x64, TILE_DIRECT = 0, 1, 2, 3, 4, 5, 99
TILE2_128x128, TILE2_128x64, TILE2_256x32, TILE2_256x16, TILE2_64x64 = 11, 12, 13, 14, 15
TILE_HALO = 20
TILE3_288 = 30
MAX_PHASES = 8

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class ConvPhase(C.Structure):
    _fields_ = [("weight", vp), ("taps", vp), ("ntaps", i32), ("out_off", i32 * 3), ("weight_frag", vp), ("weight_frag_layout", i32)]


class ConvDesc(C.Structure):
    _fields_ = [("dtype", i32), ("N", i32), ("D", i32), ("H", i32), ("W", i32), ("Cin", i32),
                ("Do", i32), ("Ho", i32), ("Wo", i32), ("stride", i32 * 3), ("pad", i32 * 3),
                ("OD", i32), ("OH", i32), ("OW", i32), ("out_stride", i32 * 3),
                ("Cout", i32), ("ldc", i32), ("cout_pad", i32), ("k_pad", i32),
                ("nphase", i32), ("flags", i32), ("tile", i32), ("stages", i32),
                ("phase", ConvPhase * MAX_PHASES)]


class ConvSkip(C.Structure):
    _fields_ = [("x", vp), ("cin", i32), ("weight_frag", vp)]


class ConvCat2(C.Structure):
    _fields_ = [("x", vp), ("cin", i32), ("H", i32), ("W", i32), ("stride", i32)]


PWCHAIN_MAX = 3


class PwChainDesc(C.Structure):
    _fields_ = [("dtype", i32), ("nlayers", i32), ("rows", i64), ("cin", i32), ("ldy", i32),
                ("cout", i32 * PWCHAIN_MAX), ("k_pad", i32 * PWCHAIN_MAX), ("flags", i32 * PWCHAIN_MAX),
                ("weight", vp * PWCHAIN_MAX), ("bias", vp * PWCHAIN_MAX), ("scale", vp * PWCHAIN_MAX), ("shift", vp * PWCHAIN_MAX),
                ("plane", i64)]


class StemDesc(C.Structure):
    _fields_ = [("dtype", i32), ("N", i32), ("H", i32), ("W", i32), ("Cin", i32), ("Cout", i32),
                ("weight", vp), ("bias", vp), ("scale", vp), ("shift", vp), ("x_layout", i32)]


class BneckDesc(C.Structure):
    _fields_ = [("dtype", i32), ("N", i32), ("H", i32), ("W", i32), ("C", i32), ("P", i32),
                ("weight", vp * 3), ("bias", vp * 3), ("scale", vp * 3), ("shift", vp * 3)]


class BneckDsDesc(C.Structure):
    _fields_ = [("dtype", i32), ("N", i32), ("H", i32), ("W", i32), ("Cin", i32), ("P", i32), ("C", i32),
                ("weight", vp * 4), ("scale", vp * 4), ("shift", vp * 4)]


class XrDesc(C.Structure):
    _fields_ = [("dtype", i32), ("C", i32), ("P", i32), ("M", C.c_int64), ("weight", vp * 2), ("scale", vp * 2), ("shift", vp * 2), ("consts", vp)]


class NamedTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", vp), ("ndim", i32), ("shape", C.c_int64 * 5)]


class VolPlanConfig(C.Structure):
    _fields_ = [("dtype", i32), ("num_layers", i32), ("style_caffe", i32), ("num_joints", i32), ("B", i32), ("NV", i32), ("H", i32), ("W", i32),
                ("volume_size", i32), ("cuboid_side", C.c_double), ("volume_multiplier", C.c_double), ("volume_softmax", i32), ("aggregation", i32),
                ("transfer_cmu_to_human36m", i32), ("use_graph", i32)]


class PlanInfo(C.Structure):
    _fields_ = [("launches", i32), ("heatmap_h", i32), ("heatmap_w", i32), ("flops", C.c_double), ("bytes_allocated", C.c_int64),
                ("n_expand_reduce", i32), ("n_bottleneck", i32), ("n_bottleneck_ds", i32), ("n_conv_cat2", i32), ("n_conv2d_halo", i32), ("n_pwchain", i32),
                ("n_stem_pool", i32), ("n_splitk", i32), ("n_conv_skip", i32), ("graph_captured", i32), ("logits", vp), ("logits_planar", i32)]


# symbol -> (restype, argtypes); must list every symbol include/lt_hip.h declares
SIGNATURES = {
    "lt_plan_create_vol": (C.c_int, [C.POINTER(VolPlanConfig), C.POINTER(NamedTensor), i32, C.POINTER(vp)]),
    "lt_plan_forward_vol": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "lt_plan_info": (C.c_int, [vp, C.POINTER(PlanInfo)]),
    "lt_plan_destroy": (None, [vp]),
    "lt_last_error": (C.c_char_p, []),
    "lt_abi_version": (C.c_int, []),
    "lt_device_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    "lt_conv_fwd": (C.c_int, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp]),
    "lt_conv_skip_fwd": (C.c_int, [C.POINTER(ConvDesc), vp, vp, vp, vp, C.POINTER(ConvSkip), vp, vp]),
    "lt_conv_cat2_fwd": (C.c_int, [C.POINTER(ConvDesc), vp, C.POINTER(ConvCat2), vp, vp, vp, vp, vp, vp]),
    "lt_conv_cout_pad": (C.c_int, [i32]),
    "lt_conv_chunk_samples": (i32, [i32, i64]),
    "lt_conv_pack_weights": (C.c_int, [vp, i32, i32, vp, vp]),
    "lt_conv_pack_weights_t32": (C.c_int, [vp, i32, i32, i32, i32, vp, vp]),
    "lt_conv_pack_weights32": (C.c_int, [vp, i32, i32, vp, vp]),
    "lt_pwchain_fwd": (C.c_int, [C.POINTER(PwChainDesc), vp, vp, vp]),
    "lt_stem_pool_fwd": (C.c_int, [C.POINTER(StemDesc), vp, vp, vp]),
    "lt_bottleneck_fwd": (C.c_int, [C.POINTER(BneckDesc), vp, vp, vp]),
    "lt_bottleneck_ds_fwd": (C.c_int, [C.POINTER(BneckDsDesc), vp, vp, vp]),
    "lt_expand_reduce_fwd": (C.c_int, [C.POINTER(XrDesc), vp, vp, vp, vp, vp]),
    "lt_stem_packed_bytes": (C.c_size_t, []),
    "lt_stem_pack_weights": (C.c_int, [vp, i32, vp, vp]),
    "lt_maxpool_fwd": (C.c_int, [i32, vp, vp, i32, i32, i32, i32, i32, i32 * 3, i32 * 3, i32 * 3, vp]),
    "lt_global_avgpool": (C.c_int, [i32, vp, vp, i32, i32, i32, vp]),
    "lt_nchw_to_nhwc": (C.c_int, [i32, vp, vp, i32, i32, i32, i32, vp]),
    "lt_nhwc_to_nchw_f32": (C.c_int, [i32, vp, vp, i32, i32, i32, i32, vp]),
    "lt_coord_volumes": (C.c_int, [vp, vp, vp, f32, i32, i32, i32, vp, vp]),
    "lt_rotate_points": (C.c_int, [vp, vp, vp, i64, vp]),
    "lt_unproject_fwd": (C.c_int, [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "lt_unproject_grid_fwd": (C.c_int, [i32, vp, vp, vp, vp, vp, f32, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "lt_softargmax3d_workspace": (C.c_size_t, [i32, i32, i64]),
    "lt_softargmax3d_fwd": (C.c_int, [vp, vp, f32, i32, i32, i32, vp, vp, i32, i32, i64, vp, vp]),
    "lt_softargmax2d_fwd": (C.c_int, [vp, f32, i32, vp, vp, i32, i32, i32, vp]),
    "lt_triangulate_dlt": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "lt_bn_act_fwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, i32, vp]),
    "lt_bn_act_bwd_workspace": (C.c_size_t, [i64, i32]),
    "lt_bn_act_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i64, i32, f32, i32, vp, vp]),
    "lt_act_bwd": (C.c_int, [vp, vp, vp, vp, vp, i32, i64, i32, vp]),
    "lt_channel_sum_workspace": (C.c_size_t, [i64, i32]),
    "lt_channel_sum": (C.c_int, [vp, i64, i32, vp, i32, vp, vp]),
    "lt_channel_sum_dt": (C.c_int, [i32, vp, i64, i32, vp, i32, vp, vp]),
    "lt_maxpool_bwd_dt": (C.c_int, [i32, vp, vp, vp, i32, i32, i32, i32, i32, i32 * 3, i32 * 3, i32 * 3, vp]),
    "lt_convert_pad": (C.c_int, [i32, vp, i32, vp, i64, i32, i32, vp]),
    "lt_maxpool_bwd": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, i32 * 3, i32 * 3, i32 * 3, vp]),
    "lt_adam_step_multi": (C.c_int, [vp, i32, i32, f32, f32, f32, f32, i32, vp]),
    "lt_add_f32": (C.c_int, [vp, vp, i64, vp]),
    "lt_pad_channels_f32": (C.c_int, [vp, vp, i64, i32, i32, vp]),
    "lt_zero": (C.c_int, [vp, i64, vp]),
    "lt_softargmax2d_bwd": (C.c_int, [vp, vp, vp, C.c_float, i32, vp, i32, i32, i32, vp]),
    "lt_triangulate_dlt_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "lt_splitk_reduce": (C.c_int, [i32, vp, i32, i64, i64, i32, vp, vp, vp, vp, vp, i32, vp]),
    "lt_add_i64_multi": (C.c_int, [vp, i32, i64, vp]),
    "lt_cast_f32_bf16": (C.c_int, [vp, vp, i64, vp]),
    "lt_gather_f32_multi": (C.c_int, [vp, i32, i32, vp]),
    "lt_amax_f32": (C.c_int, [vp, i64, vp, vp]),
    "lt_amax_dt": (C.c_int, [i32, vp, i64, vp, vp]),
    "lt_quant_fp8_dt": (C.c_int, [i32, vp, vp, i64, vp, vp, vp]),
    "lt_quant_fp8": (C.c_int, [vp, vp, i64, vp, vp, vp]),
    "lt_gather_f32_fp8": (C.c_int, [vp, vp, vp, i64, vp, vp, vp]),
    "lt_scale_product": (C.c_int, [vp, i32, vp, vp, vp]),
    "lt_gather_f32": (C.c_int, [vp, vp, vp, i64, vp]),
    "lt_gather_f32_bf16": (C.c_int, [vp, vp, vp, i64, vp]),
    "lt_conv_wgrad_workspace": (C.c_size_t, [i64, i32, i32]),
    "lt_pack_n8_bf16_bytes": (C.c_size_t, [i32, i64, i32]),
    "lt_pack_n8_bf16": (C.c_int, [vp, vp, i32, i64, i32, i32, vp]),
    "lt_pack_n8_from_bf16": (C.c_int, [vp, vp, i32, i64, i32, i32, vp]),
    "lt_conv_wgrad_bf16_workspace": (C.c_size_t, [i64, i32, i32]),
    "lt_conv_wgrad_bf16": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32 * 3, i32 * 3, i32, i32, i32, i32, i32, i32, vp, vp]),
    "lt_conv_wgrad_bf16_nhwc_ok": (C.c_int, [i32, i32, i32, i32, i32, i32, i32, i32, i32, i32 * 3, i32 * 3, i32, i32, i32, i32, i32]),
    "lt_conv_wgrad_bf16_nhwc": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32 * 3, i32 * 3, i32, i32, i32, i32, i32, i32, vp, vp]),
    "lt_conv_wgrad": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32 * 3, i32 * 3, i32, i32, i32, i32, i32, i32, vp, vp]),
    "lt_adam_step": (C.c_int, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp]),
    "lt_unproject_bwd_workspace": (C.c_size_t, [i32, i32, i32, i32, i32, i32]),
    "lt_unproject_bwd": (C.c_int, [i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, C.c_size_t, vp]),
    "lt_global_avgpool_bwd": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
    "lt_global_avgpool_bwd_dt": (C.c_int, [i32, vp, vp, i32, i32, i32, i32, vp]),
    "lt_softargmax3d_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, f32, i32, i32, vp, i32, i32, i64, vp]),
    "lt_softargmax3d_bwd_dense": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, f32, i32, i32, vp, i32, i32, i64, vp]),
    "lt_volumetric_ce_fwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i64, vp]),
    "lt_bn_stats_workspace": (C.c_size_t, [i64, i32]),
    "lt_bn_stats_fwd": (C.c_int, [i32, vp, i64, i32, vp, vp, vp, vp, f32, vp, vp]),
    "lt_graph_begin": (C.c_int, [vp]),
    "lt_graph_end": (C.c_int, [vp, C.POINTER(vp)]),
    "lt_graph_launch": (C.c_int, [vp, vp]),
    "lt_graph_destroy": (C.c_int, [vp]),
    "lt_event_create": (C.c_int, [C.POINTER(vp)]),
    "lt_event_record": (C.c_int, [vp, vp]),
    "lt_event_elapsed_ms": (C.c_int, [vp, vp, C.POINTER(f32)]),
    "lt_event_destroy": (C.c_int, [vp]),
}

_lib = None


def lib():
    """The loaded library (loads on first use; raises if it was never built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("liblt_hip.so is not built (%s): run `python __graft_entry__.py` (build()) first; "
                               "there is no non-HIP fallback" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        if l.lt_abi_version() != 1:
            raise RuntimeError("liblt_hip.so ABI version mismatch")
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().lt_last_error()
        raise RuntimeError("liblt_hip %s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def dtype_code(dt):
    if dt == torch.float32:
        return LT_F32
    if dt == torch.bfloat16:
        return LT_BF16
    if dt == torch.float8_e4m3fn:          # operands of lt_conv_fwd only (the fp8 V2V convolutions of the training step)
        return LT_FP8
    raise TypeError("liblt_hip supports float32 and bfloat16 activations (and float8_e4m3fn convolution operands), got %s" % dt)


def require_gpu(t, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU: this package has no CPU path (device=%s)" % (name, t.device))


def cur_stream():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()


def i3(a, b=None, c=None):
    if b is None:
        a, b, c = a
    return (i32 * 3)(int(a), int(b), int(c))


def device_info():
    cu, lds = C.c_int(0), C.c_int(0)
    buf = C.create_string_buffer(64)
    check(lib().lt_device_info(C.byref(cu), C.byref(lds), buf, 64), "lt_device_info")
    return {"cu_count": cu.value, "lds_per_cu": lds.value, "arch": buf.value.decode()}


class Event:
    """hipEvent on an explicit stream (torch.cuda.Event only sees torch's current stream)."""

    def __init__(self):
        h = vp()
        check(lib().lt_event_create(C.byref(h)), "lt_event_create")
        self.h = h

    def record(self, stream=None):
        check(lib().lt_event_record(self.h, cur_stream() if stream is None else stream), "lt_event_record")

    def elapsed_ms(self, stop):
        ms = f32(0)
        check(lib().lt_event_elapsed_ms(self.h, stop.h, C.byref(ms)), "lt_event_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            lib().lt_event_destroy(self.h)
        except Exception:
            pass


class Graph:
    """A captured hipGraph of liblt_hip launches."""

    def __init__(self):
        self.exec = None

    def capture(self, stream, fn):
        check(lib().lt_graph_begin(stream), "lt_graph_begin")
        try:
            fn()
        finally:
            h = vp()
            rc = lib().lt_graph_end(stream, C.byref(h))
        check(rc, "lt_graph_end")
        self.exec = h

    def launch(self, stream):
        check(lib().lt_graph_launch(self.exec, stream), "lt_graph_launch")

    def __del__(self):
        try:
            if self.exec is not None:
                lib().lt_graph_destroy(self.exec)
        except Exception:
            pass
