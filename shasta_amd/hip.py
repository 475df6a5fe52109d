"""ctypes binding of the C ABI in include/shasta_hip.h (libshasta_hip.so, built in-tree by shasta_amd.build).

PyTorch is only plumbing here: it owns device memory and the current HIP stream; every compute call goes
through the C ABI with raw device pointers.  There is NO fallback: if the library is missing or a call fails,
`ShastaHipError` is raised.
"""
import ctypes as C
import os

import torch

ABI_VERSION = 15  # must equal shasta_abi_version() of the loaded library
# SHASTA_HIP_LIB: load another build of the same ABI (A/B timing of kernel variants on one box)
_LIB_PATH = os.environ.get("SHASTA_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libshasta_hip.so")
_lib = None


class ShastaHipError(RuntimeError):
    pass


E_UNSUPPORTED = -5  # include/shasta_hip.h SHASTA_E_UNSUPPORTED
OPT_F32_WEIGHT_STREAM = 1  # include/shasta_hip.h SHASTA_OPT_*
OPT_F32_EMBED_GEMM = 2
OPT_F32_AFF = 4
OPT_F16X2_WEIGHT_STREAM = 8
OPT_F16X2_PAIR = 16
OPT_PRECUT_WEIGHT_STREAM = 32
OPT_F16GRID_PAIR = 64
OPT_TWO_PASS_AFF = 128
OPT_F16X2_AFF = 256
OPT_ONE_PASS_AFF = 512
F16X2_MAX_ROW_RATIO = 16384.0  # SHASTA_F16X2_MAX_ROW_RATIO
PRECUT_MIN_BATCH = 17  # csrc/anchor.hip: from this many frame-pairs per call the pre-cut fp16 weight stream serves the call


class Linear(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p)]


class Weights(C.Structure):
    _fields_ = [("max_obj", C.c_int), ("num_feats", C.c_int), ("feat_dim", C.c_int), ("options", C.c_int),
                ("aug_shape", (Linear * 2) * 4), ("aug_dets", (Linear * 2) * 4),
                ("fuse_shape", Linear * 4), ("fuse_det", Linear * 3), ("res_coeff", Linear * 3),
                ("aff", Linear * 6), ("aug_shape_aux", C.c_void_p), ("aug_shape_aux_bytes", C.c_size_t)]


# every symbol include/shasta_hip.h declares: name -> (restype, argtypes)
_P, _I, _F, _Z = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_WP = C.POINTER(Weights)
SYMBOLS = {
    "shasta_abi_version": (_I, []),
    "shasta_build_info": (C.c_char_p, []),
    "shasta_last_error": (C.c_char_p, []),
    "shasta_voxelize_workspace_bytes": (_Z, [_I, _I, _I]),
    "shasta_voxelize_mean_f32": (_I, [_P, _I, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "shasta_voxelize_batch_workspace_bytes": (_Z, [_P, _I, _I, _I]),
    "shasta_voxelize_mean_batch_f32": (_I, [_P, _P, _I, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "shasta_voxel_mean_f32": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "shasta_bev_gather_f32": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _F, _F, _F, _F, _F, _P, _I, _I, _P]),
    "shasta_shared_conv_packed_bytes": (_Z, [_I]),
    "shasta_shared_conv_pack_f32": (_I, [_P, _P, _P, _P, _P, _P, _F, _I, _P, _Z, _P]),
    "shasta_shared_conv_f32": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "shasta_shared_conv_f16x2_supported": (_I, [_I, _I, _I]),
    "shasta_shared_conv_f16x2_packed_bytes": (_Z, [_I]),
    "shasta_shared_conv_pack_f16x2": (_I, [_P, _P, _P, _P, _P, _P, _F, _I, _P, _Z, _P]),
    "shasta_shared_conv_multi_workspace_bytes": (_Z, [_I]),
    "shasta_shared_conv_multi_workspace_bytes_for": (_Z, [_I, _I, _I, _I, _I, _I]),
    "shasta_shared_conv_multi_f32": (_I, [_P, _P, _I, _I, _I, _I, _P, _Z, _I, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _P, _Z, _P]),
    "shasta_shared_conv_multi_bounded_f32": (_I, [_P, _P, _I, _I, _I, _I, _P, _Z, _I, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _P, _Z, _F, _P]),
    "shasta_conv_train_supported": (_I, [_I, _I, _I]),
    "shasta_shared_conv_pack_raw_f32": (_I, [_P, _P, _I, _P, _Z, _P]),
    "shasta_shared_conv_pack_raw_f16x2": (_I, [_P, _P, _I, _P, _Z, _P]),
    "shasta_bn_workspace_bytes": (_Z, []),
    "shasta_bn_stats_f32": (_I, [_P, C.c_long, _P, _P, _Z, _P]),
    "shasta_bn_finalize_f32": (_I, [_P, C.c_double, _F, _F, _P, _P, _P, _P, _P]),
    "shasta_bn_relu_apply_f32": (_I, [_P, C.c_long, _P, _P, _P, _P, _P]),
    "shasta_bn_relu_bwd_reduce_f32": (_I, [_P, _P, C.c_long, _P, _P, _P, _P, _P, _Z, _P]),
    "shasta_conv_dy_bytes": (_Z, [_I, _I, _I]),
    "shasta_bn_relu_bwd_dy_f16x2": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, C.c_double, _P, _Z, _P, _P, _I, _P]),
    "shasta_conv_wgrad_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "shasta_conv_wgrad_f16x2": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "shasta_packed_bytes": (_Z, [_I, _I, _I]),
    "shasta_pack_weights_f32": (_I, [_WP, _P, _Z, _P]),
    "shasta_aug_shape_aux_bytes": (_Z, [_I, _I, _I]),
    "shasta_aug_shape_aux_f32": (_I, [_WP, _P, _Z, _P]),
    "shasta_aug_shape_aux_row_ratio": (_I, [_I, _I, _P, _Z, C.POINTER(C.c_float), C.POINTER(C.c_int), _P]),
    "shasta_forward_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "shasta_affinity_forward_f32": (_I, [_WP, _P, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "shasta_affinity_from_bev_f32": (_I, [_WP, _P, _I, _P, _P, _I, _I, _I, _F, _F, _F, _F, _F, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _Z,
                                          _P, C.POINTER(C.c_void_p)]),
    "shasta_affinity_forward_train_f32": (_I, [_WP, _P, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "shasta_affinity_forward_timed_f32": (_I, [_WP, _P, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _Z, _P, _P, _P, _P, _P]),
    "shasta_event_create": (_I, [C.POINTER(C.c_void_p)]),
    "shasta_event_destroy": (_I, [_P]),
    "shasta_event_elapsed_ms": (_I, [_P, _P, C.POINTER(C.c_float)]),
    "shasta_anchor_shape_f32": (_I, [_WP, _I, _P, _P, _P, _Z, _P]),
    "shasta_anchor_boxes_f32": (_I, [_WP, _I, _P, _P, _I, _P, _P, _P, _Z, _P]),
    "shasta_pair_residual_f32": (_I, [_WP, _P, _I, _P, _P, _P, _P, _P, _I, _P, _Z, _P]),
    "shasta_aff_softmax_f32": (_I, [_WP, _P, _I, _P, _I, _P, _P, _P, _P, _Z, _P]),
    "shasta_aff_status": (_I, [_WP, _I, _I, _P, _Z, C.POINTER(C.c_int), _P]),
    "shasta_forward_status": (_I, [_WP, _I, _P, _Z, C.POINTER(C.c_int), _P]),
    "shasta_iou3d_distance_f64": (_I, [_P, _I, _P, _I, _I, _I, _P, _P]),
    "shasta_hand_dist_f32": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _P, _P]),
    "shasta_hand_dist_bwd_f32": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "shasta_combine_bwd_f32": (_I, [_P, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "shasta_softmax_bwd_f32": (_I, [_P, _P, _P, _P, _I, _I, _P, _I, _P]),
    "shasta_pair_hidden_f32": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P]),
    "shasta_pair_reduce_f32": (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "shasta_pair_mlp_supported": (_I, [_I]),
    "shasta_pair_mlp_grad_floats": (_I, [_I, _I]),
    "shasta_pair_mlp_workspace_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "shasta_pair_mlp_forward_f32": (_I, [_I, _I, _P, _P, C.POINTER(_P), _I, _I, _I, _P, _P]),
    "shasta_pair_mlp_backward_f32": (_I, [_I, _I, _P, _P, C.POINTER(_P), _P, _I, _I, _I, _P, _P, _P, _P, _Z, _P]),
    "shasta_colsum_f32": (_I, [_P, _I, _I, _I, _P, _P, C.c_size_t, _P]),
    "shasta_lowrank_outer_f32": (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _P]),
    "shasta_smallm_nn_workspace_bytes": (_Z, [_I, _I, _I]),
    "shasta_smallm_nn_f32": (_I, [_P, _I, _P, _I, _I, _I, _P, C.c_long, _I, _P, _Z, _P]),
    "shasta_scale_f32": (_I, [_P, C.c_long, _F, _P]),
    "shasta_adam_prepare_f32": (_I, [_P, _P, _P, _P]),
    "shasta_adam_step_f32": (_I, [_P, _P, _P, _P, C.c_long, _F, _F, _F, _F, _F, _I, _P, _P]),
    "shasta_affinity_loss_f32": (_I, [_P, _P, _P, _I, _I, _P, _P, _P]),
    "shasta_affinity_loss_bwd_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P]),
    "shasta_adam_multi_f32": (_I, [_I, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_long), _F, _F, _F, _F, _F, _I, _P, _P]),
    "shasta_adam_lowrank_dx_workspace_bytes": (_Z, [_I, _I, _I]),
    "shasta_adam_lowrank_dx_f32": (_I, [_P, _P, _P, _I, _I, _P, _I, _P, _I, _I, _P, _I, _I, _P, C.c_long, _I, _P, _Z, _F, _F, _F, _F, _F, _I, _P, _P]),
    "shasta_adam_lowrank_f32": (_I, [_P, _P, _P, _I, _I, _P, _I, _P, _I, _I, _F, _F, _F, _F, _F, _I, _P, _P]),
    "shasta_abs_f32": (_I, [_P, _P, _P, C.c_long, _I, _I, _I, _I, _P]),
    "shasta_bev_gather_bwd_f32": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _F, _F, _F, _F, _F, _I, _I, _P, _P]),
    "shasta_nms_workspace_bytes": (_Z, [_I]),
    "shasta_nms_rotated_f32": (_I, [_P, _I, _F, _P, _Z, _P, _P, _P]),
    "shasta_nms_normal_f32": (_I, [_P, _I, _F, _P, _Z, _P, _P, _P]),
    "shasta_boxes_bev_f32": (_I, [_P, _I, _P, _I, _I, _P, _P]),
    "shasta_center_greedy_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P]),
    "shasta_track_merged_f64": (_I, [_P] * 9 + [_I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "shasta_decode_flags_f32": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "shasta_gemm_strided_f32": (_I, [_P, C.c_long, C.c_long, _P, C.c_long, C.c_long, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "shasta_gemm_strided_group_f32": (_I, [_I, _P, _P, _P, _P, _P, C.c_long, C.c_long, C.c_long, C.c_long, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "shasta_gemm_nt_f32": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "shasta_gemm_nt_pieces_f32": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P]),
}


def lib_path():
    return _LIB_PATH


def load():
    """Load libshasta_hip.so and bind every declared symbol.  Raises ShastaHipError when absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ShastaHipError("HIP extension missing: %s (run `python -m shasta_amd.build`); "
                             "there is no CPU fallback" % _LIB_PATH)
    lib = C.CDLL(_LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.shasta_abi_version() != ABI_VERSION:
        raise ShastaHipError("%s has ABI version %d, this binding needs %d: rebuild with `python -m shasta_amd.build`"
                             % (_LIB_PATH, lib.shasta_abi_version(), ABI_VERSION))
    if not os.environ.get("SHASTA_HIP_LIB"):  # (a variant library for an A/B is built from other sources on purpose)
        from . import build as _build
        info, want = lib.shasta_build_info().decode(), _build.source_hash()
        if not info.endswith("src " + want):
            raise ShastaHipError("%s was built from other sources than the ones next to it (%s, sources now %s): the last build failed or "
                                 "was forgotten - run `python -m shasta_amd.build`" % (_LIB_PATH, info.rsplit(" ", 1)[-1], want))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise ShastaHipError("%s failed (%d): %s" % (what, rc, load().shasta_last_error().decode()))


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise ShastaHipError("expected a device tensor (the HIP path has no CPU fallback)")
    if t.dtype not in (torch.float32, torch.float64, torch.int32, torch.uint8):
        raise ShastaHipError("unexpected dtype %s" % t.dtype)
    if not t.is_contiguous():
        raise ShastaHipError("expected a contiguous tensor")
    return C.c_void_p(t.data_ptr())


def ptr_view(t):
    """Pointer to the first element of a strided VIEW (column block of a matrix): the caller passes the strides."""
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.float32:
        raise ShastaHipError("expected an fp32 device tensor (the HIP path has no CPU fallback)")
    return C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr():
    """The current HIP stream of the current device as the C ABI takes it.  (torch.cuda.current_stream() costs ~9 us of Python per call -
    1.1 ms of a training step's ~120 launches - so the raw binding underneath it is used where this torch has it.)"""
    if _raw_stream is not None and _cur_device is not None:
        try:
            return C.c_void_p(_raw_stream(_cur_device()))
        except RuntimeError:  # (the runtime not initialised yet: the public call below initialises it)
            pass
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
