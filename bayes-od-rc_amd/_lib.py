"""ctypes binding of libbayesod_hip.so (include/bayesod.h).  cffi is not installed in the target
image; the ABI is plain C so a cffi ABI-mode binding is the same declarations."""
import ctypes as C
import os

import numpy as np

# (GPU_MAX_HW_QUEUES: a handle drives up to five HIP streams -- a training handle seven -- which the runtime multiplexes onto that many
# hardware queues (default 4).  Only the training step gains from 8 (docs/DESIGN_HISTORY.md A.1), and the variable changes queue
# multiplexing for every HIP user of the process: it is a default of the ENTRY POINTS that benefit -- run_training.py, bench.py,
# which records the effective value in its JSON line -- not of importing this package.)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libbayesod_hip.so")
# development aid (tests/tools/ab_lib.sh): same-box A/B of two BUILDS of the library -- the override must name an existing file
if os.environ.get("BOD_LIB_OVERRIDE"):
    LIB_PATH = os.path.abspath(os.environ["BOD_LIB_OVERRIDE"])

BOD_OK, BOD_ERR_INVALID_ARG, BOD_ERR_HIP, BOD_ERR_OOM, BOD_ERR_NOT_READY, BOD_ERR_NO_DEVICE = range(6)


class BodConfig(C.Structure):
    _fields_ = [
        ("device", C.c_int32), ("image_h", C.c_int32), ("image_w", C.c_int32), ("batch", C.c_int32),
        ("mc_samples", C.c_int32), ("num_classes", C.c_int32), ("anchors_per_location", C.c_int32),
        ("min_level", C.c_int32), ("max_level", C.c_int32), ("dropout_rate", C.c_float),
        ("use_full_covar", C.c_int32), ("dirichlet_non_informative", C.c_int32),
        ("gaussian_isotropic", C.c_int32), ("isotropic_variance", C.c_float),
        ("ranking_method", C.c_int32), ("nms_max_output_size", C.c_int32),
        ("nms_iou_threshold", C.c_float), ("nms_soft_sigma", C.c_float), ("nms_variant", C.c_int32),
        ("num_categorical_draws", C.c_int32), ("has_covar_head", C.c_int32),
        ("kitti_scale_h", C.c_float), ("kitti_scale_w", C.c_float), ("precision", C.c_int32),
        ("mc_sample_base", C.c_int32),
        ("mc_ensemble_size", C.c_int32),
        ("training", C.c_int32),
        ("backbone_depth", C.c_int32),
        ("pipeline_overlap", C.c_int32),
        ("reserved", C.c_int32 * 2),
    ]


class BodSizes(C.Structure):
    _fields_ = [
        ("num_pixels", C.c_int32), ("num_anchors", C.c_int32), ("level_h", C.c_int32 * 8),
        ("level_w", C.c_int32 * 8), ("num_levels", C.c_int32), ("max_detections", C.c_int32),
        ("device_bytes", C.c_int64),
    ]


_F = C.POINTER(C.c_float)
_I = C.POINTER(C.c_int32)
_H = C.c_void_p

# name -> (restype, argtypes); must list every symbol include/bayesod.h declares
SIGNATURES = {
    "bod_version": (C.c_char_p, []),
    "bod_last_error": (C.c_char_p, [_H]),
    "bod_create": (C.c_int, [C.POINTER(BodConfig), C.POINTER(_H)]),
    "bod_destroy": (C.c_int, [_H]),
    "bod_query_sizes": (C.c_int, [_H, C.POINTER(BodSizes)]),
    "bod_update_config": (C.c_int, [_H, C.POINTER(BodConfig)]),
    "bod_load_weight": (C.c_int, [_H, C.c_char_p, C.c_int32, C.POINTER(C.c_int64), C.c_int32, _F]),
    "bod_finalize_weights": (C.c_int, [_H]),
    "bod_set_anchors": (C.c_int, [_H, _F, C.c_int32]),
    "bod_forward": (C.c_int, [_H, C.c_void_p, C.c_int32, C.c_uint64, C.c_uint32]),
    "bod_get_raw": (C.c_int, [_H, _F, _F, _F]),
    "bod_set_raw": (C.c_int, [_H, _F, _F, _F]),
    "bod_get_pyramid": (C.c_int, [_H, C.c_int32, _F]),
    "bod_posterior": (C.c_int, [_H, C.c_uint64, C.c_uint32]),
    "bod_validation_post": (C.c_int, [_H]),
    "bod_get_num_kept": (C.c_int, [_H, _I]),
    "bod_get_posterior": (C.c_int, [_H, C.c_int32, _F, _F, _F, _F, _F, _I]),
    "bod_set_posterior": (C.c_int, [_H, C.c_int32, C.c_int32, _F, _F, _F, _F]),
    "bod_nms": (C.c_int, [_H]),
    "bod_get_nms": (C.c_int, [_H, C.c_int32, _I, _I]),
    "bod_set_nms": (C.c_int, [_H, C.c_int32, _I, C.c_int32]),
    "bod_get_iou_matrix": (C.c_int, [_H, C.c_int32, _F]),
    "bod_cluster_fuse": (C.c_int, [_H]),
    "bod_set_affinity": (C.c_int, [_H, C.c_int32, _F, C.c_int32, C.c_int32]),
    "bod_get_detections": (C.c_int, [_H, C.c_int32, _I, _F, _F, _F, _F]),
    "bod_get_detections_batch": (C.c_int, [_H, _I, _F, _F, _F, _F]),
    "bod_device_detections": (C.c_int, [_H, C.c_int32, C.POINTER(C.c_void_p)]),
    "bod_device_raw": (C.c_int, [_H, C.POINTER(C.c_void_p), C.c_int32]),
    "bod_infer": (C.c_int, [_H, C.c_void_p, C.c_int32, C.c_uint64, C.c_uint32]),
    "bod_infer_async": (C.c_int, [_H, C.c_void_p, C.c_int32, C.c_uint64, C.c_uint32, _I]),
    "bod_collect": (C.c_int, [_H, C.c_int32, _I, _F, _F, _F, _F]),
    "bod_upload_images": (C.c_int, [_H, _F]),
    "bod_upload_frames_u8": (C.c_int, [_H, C.POINTER(C.c_uint8), C.c_int32, C.c_int32, _F, C.c_int32]),
    "bod_upload_frames_u8_async": (C.c_int, [_H, C.POINTER(C.c_uint8), C.c_int32, C.c_int32, _F, C.c_int32, C.c_int32]),
    "bod_device_images_buffer": (C.c_void_p, [_H, C.c_int32]),
    "bod_device_images": (C.c_void_p, [_H]),
    "bod_synchronize": (C.c_int, [_H]),
    "bod_stage_conv_wgrad": (C.c_int, [C.c_int32, _F, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _F, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_int32, C.c_int32, _F, _F]),
    "bod_stage_conv": (C.c_int, [C.c_int32, _F, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _F, _F,
                                 C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _F,
                                 C.c_float, C.c_uint64, C.c_int32, C.c_uint32, C.c_int32, C.c_int32, _F]),
    "bod_loss_forward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _F, _F, _F, _F, _F, _F,
                                   C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_int32, C.c_int32, C.c_float,
                                   C.POINTER(C.c_double)]),
    "bod_train_step": (C.c_int, [_H, C.c_void_p, C.c_int32, _F, _F, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_uint64, C.c_uint32,
                                 C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, C.POINTER(C.c_double)]),
    "bod_train_gradients": (C.c_int, [_H, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "bod_train_apply": (C.c_int, [_H, C.c_float, C.POINTER(C.c_double)]),
    "bod_train_get": (C.c_int, [_H, C.c_char_p, C.c_int32, C.c_int32, _F, C.c_int64]),
    "bod_train_set": (C.c_int, [_H, C.c_char_p, C.c_int32, C.c_int32, _F, C.c_int64]),
    "bod_train_step_count": (C.c_int, [_H, C.POINTER(C.c_int64), C.c_int64]),
    "bod_loss_backward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _F, _F, _F, _F, _F, _F,
                                    C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_int32, C.c_int32, C.c_float,
                                    C.c_float, C.c_float, C.POINTER(C.c_double), _F, _F, _F]),
    "bod_bench_head_conv": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "bod_profile_begin": (C.c_int, [_H]),
    "bod_profile_select": (C.c_int, [_H, C.c_int32]),
    "bod_plan_info": (C.c_int, [_H, C.POINTER(C.c_int32)]),
    "bod_record_width": (C.c_int32, [_H]),
    "bod_gather_detections": (C.c_int, [_H, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "bod_profile_end": (C.c_int, [_H, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                  C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}

_lib = None


def load():
    """Loads the shared library; raises RuntimeError (never falls back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libbayesod_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError => header / library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def fptr(a):
    return None if a is None else a.ctypes.data_as(_F)


def iptr(a):
    return None if a is None else a.ctypes.data_as(_I)


def check(lib, handle, status):
    if status == BOD_OK:
        return
    msg = lib.bod_last_error(handle)
    msg = msg.decode() if msg else "status %d" % status
    if status in (BOD_ERR_INVALID_ARG, BOD_ERR_NOT_READY):
        raise ValueError(msg)
    if status == BOD_ERR_OOM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def as_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)
