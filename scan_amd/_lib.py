"""ctypes binding of libscan_hip.so (the C ABI declared in include/scan_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  ``lib()``
raises if the shared object is missing and every compute entry point raises
``RuntimeError`` (with ``scan_last_error()``) on a non-zero return, mirroring
the reference's AT_ASSERTM/AT_ERROR -> RuntimeError behaviour
(reference fcos_core/csrc/nms.h:10-28, csrc/SigmoidFocalLoss.h:10-41).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SCAN_HIP_LIB: an alternative build of the same library (timing experiments, see csrc/Makefile); default = the product
LIB_PATH = os.environ.get("SCAN_HIP_LIB") or os.path.join(_HERE, "libscan_hip.so")

MAX_LEVELS = 5
TUNE_UNKNOWN = -2 ** 31  # SCAN_TUNE_UNKNOWN
NMS_PANEL = 8192     # SCAN_NMS_PANEL: single-workgroup fast path
NMS_MAX = 262144     # SCAN_NMS_MAX

c_i32, c_i64, c_f32, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p


class PyramidDesc(ctypes.Structure):
    """scan_pyramid_t"""
    _fields_ = [("n_levels", c_i32), ("n_images", c_i32), ("h", c_i32 * MAX_LEVELS), ("w", c_i32 * MAX_LEVELS),
                ("row_off", c_i64 * (MAX_LEVELS + 1))]


_PD = ctypes.POINTER(PyramidDesc)


class SgdSegment(ctypes.Structure):
    """scan_sgd_segment_t"""
    _fields_ = [("p", c_vp), ("g", c_vp), ("buf", c_vp), ("n", c_i64), ("lr", c_f32), ("wd", c_f32),
                ("first_step", c_i32), ("reserved", c_i32)]


class CkaBranch(ctypes.Structure):
    """scan_cka_branch_t"""
    _fields_ = [("w0", c_vp), ("b0", c_vp), ("w2", c_vp), ("b2", c_vp)]


SGD_MAX_SEGMENTS = 32
CKA_MAX_CLASSES = 16
SPLIT_JOB_WORDS = 11

# name -> (restype, argtypes); every symbol include/scan_hip.h declares
SIGNATURES = {
    "scan_last_error": (ctypes.c_char_p, []),
    "scan_abi_version": (ctypes.c_int, []),
    "scan_comm_standin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_double, ctypes.c_int32, ctypes.c_double,
                                         ctypes.c_void_p]),
    "scan_tune": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    "scan_take_images_backward": (ctypes.c_int, [c_vp, _PD, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "scan_cond_rnn_forward": (ctypes.c_int, [c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "scan_cond_rnn_ws_floats": (ctypes.c_int64, []),
    "scan_cond_rnn_backward": (ctypes.c_int, [c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "scan_tune_get": (ctypes.c_int, [ctypes.c_char_p]),
    "scan_tune_default": (ctypes.c_int, [ctypes.c_char_p]),
    "scan_tune_key": (ctypes.c_char_p, [ctypes.c_int]),
    "scan_mfma_sustained_bf16": (ctypes.c_int, [ctypes.c_double, c_i32, ctypes.POINTER(ctypes.c_double), c_vp]),
    "scan_conv3x3_bf16x3_instance": (ctypes.c_int, [_PD, c_i32]),
    "scan_conv1x1_bf16x6_instance": (ctypes.c_int, [_PD, c_i32, c_i32]),
    "scan_conv3x3_bf16x6_instance": (ctypes.c_int, [_PD, c_i32]),
    "scan_sigmoid_focal_loss_forward": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_f32, c_f32, c_vp, c_vp, c_vp]),
    "scan_sigmoid_focal_loss_backward": (ctypes.c_int, [c_vp, c_vp, c_vp, c_f32, c_i64, c_i32, c_f32, c_f32, c_vp, c_vp]),
    "scan_iou_loss_forward": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "scan_iou_loss_backward": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "scan_bce_logits_forward": (ctypes.c_int, [c_vp, c_vp, c_f32, c_vp, c_i64, c_i64, c_vp, c_vp]),
    "scan_bce_logits_backward": (ctypes.c_int, [c_vp, c_vp, c_f32, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "scan_cka_bce_forward": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_f32, c_vp, c_vp]),
    "scan_cka_bce_backward": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_f32, c_vp, c_vp, c_vp]),
    "scan_cka_bce_forward_loss": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_f32, c_vp, c_vp]),
    "scan_cka_bce_backward_loss": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp]),
    "scan_scale": (ctypes.c_int, [c_vp, c_f32, c_vp, c_i64, c_vp]),
    "scan_copy_cols": (ctypes.c_int, [c_vp, c_i32, c_vp, c_i32, c_i64, c_i32, c_i32, c_vp]),
    "scan_paradigm_update": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "scan_dynconv_softmax_forward": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "scan_dynconv_ws_floats": (c_i64, [c_i64, c_i32, c_i32]),
    "scan_dynconv_softmax_backward": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "scan_softmax_focal_forward": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_f32, c_vp, c_vp]),
    "scan_softmax_focal_backward": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_f32, c_f32, c_vp, c_vp]),
    "scan_nms_ws_bytes": (c_i64, [c_i64]),
    "scan_nms": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_f32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "scan_conv2d_forward": (ctypes.c_int, [c_vp, _PD, c_i32, c_vp, c_vp, c_vp, _PD, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "scan_conv2d_dgrad": (ctypes.c_int, [c_vp, _PD, c_i32, c_vp, c_vp, _PD, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "scan_conv2d_wgrad_ws_floats": (c_i64, [_PD, c_i32, c_i32, c_i32]),
    "scan_conv2d_wgrad": (ctypes.c_int, [c_vp, _PD, c_i32, c_vp, _PD, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp]),
    "scan_weight_split": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp]),
    "scan_conv3x3_bf16x3": (ctypes.c_int, [c_vp, _PD, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "scan_conv3x3_wgrad_bf16x3_ws_floats": (c_i64, [_PD, c_i32, c_i32]),
    "scan_dbscan_ws_bytes": (c_i64, [c_i64]),
    "scan_conv1x1_wgrad_bf16x3_ws_floats": (c_i64, [_PD, c_i32, c_i32]),
    "scan_conv3x3_wgrad_bf16x3": (ctypes.c_int, [c_vp, _PD, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "scan_weight_transpose": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp]),
    "scan_colsum_ws_floats": (c_i64, [c_i64, c_i32]),
    "scan_colsum": (ctypes.c_int, [c_vp, c_i64, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp]),
    "scan_relu_backward": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "scan_groupnorm_stats": (ctypes.c_int, [c_vp, _PD, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp]),
    "scan_groupnorm_relu_forward": (ctypes.c_int, [c_vp, _PD, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "scan_groupnorm_relu_forward_from_sums": (ctypes.c_int, [c_vp, _PD, c_i32, c_i32, c_vp, c_f32, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "scan_groupnorm_ws_floats": (c_i64, [_PD, c_i32, c_i32]),
    "scan_fcos_assign": (ctypes.c_int, [_PD, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "scan_fcos_compact": (ctypes.c_int, [_PD, c_vp, c_vp, c_vp, c_vp]),
    "scan_fcos_nodes_count": (c_i64, [_PD, c_vp]),
    "scan_fcos_nodes": (ctypes.c_int, [_PD, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "scan_groupnorm_relu_forward_ld": (ctypes.c_int, [c_vp, _PD, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp]),
    "scan_groupnorm_relu_forward_from_sums_ld": (ctypes.c_int, [c_vp, _PD, c_i32, c_i32, c_vp, c_f32, c_vp, c_vp, c_i32, c_vp,
                                                                c_i32, c_vp, c_vp]),
    "scan_groupnorm_relu_backward_ld": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, _PD, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp,
                                                       c_vp, c_vp, c_i32, c_vp, c_vp]),
    "scan_groupnorm_relu_backward": (ctypes.c_int, [c_vp, c_vp, c_vp, _PD, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "scan_maxpool2x2_forward": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "scan_maxpool2x2_backward": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp]),
    "scan_conv3x3_gn_bf16x3": (ctypes.c_int, [c_vp, _PD, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp]),
    "scan_conv3x3_gn_acc_bf16x3": (ctypes.c_int, [c_vp, _PD, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp]),
    "scan_groupnorm_stats_from_sums": (ctypes.c_int, [c_vp, _PD, c_i32, c_i32, c_f32, c_vp, c_vp]),
    "scan_conv3x3_pool2_bf16x3": (ctypes.c_int, [c_vp, _PD, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "scan_conv1x1_bf16x3": (ctypes.c_int, [c_vp, _PD, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, _PD, c_i32, c_i32, c_i32,
                                           c_i32, c_vp]),
    "scan_conv1x1_wgrad_bf16x3": (ctypes.c_int, [c_vp, _PD, c_i32, c_vp, _PD, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp,
                                                 c_vp]),
    "scan_conv_smallcin_bf16x3": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32,
                                                 c_i32, c_vp]),
    "scan_weight_split3": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "scan_conv3x3_bf16x6": (ctypes.c_int, [c_vp, _PD, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "scan_conv1x1_bf16x6": (ctypes.c_int, [c_vp, _PD, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, _PD, c_i32, c_i32, c_i32,
                                           c_i32, c_vp]),
    "scan_conv3x3_gn_bf16x6": (ctypes.c_int, [c_vp, _PD, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_vp, c_i32,
                                              c_vp]),
    "scan_conv3x3_pool2_bf16x6": (ctypes.c_int, [c_vp, _PD, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32,
                                                 c_vp]),
    "scan_conv3x3_wgrad_bf16x6_ws_floats": (c_i64, [_PD, c_i32, c_i32]),
    "scan_conv3x3_wgrad_bf16x6": (ctypes.c_int, [c_vp, _PD, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "scan_conv1x1_wgrad_bf16x6_ws_floats": (c_i64, [_PD, c_i32, c_i32]),
    "scan_conv1x1_wgrad_bf16x6": (ctypes.c_int, [c_vp, _PD, c_i32, c_vp, _PD, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp,
                                                 c_vp]),
    "scan_conv_smallcin_bf16x6": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32,
                                                 c_i32, c_vp]),
    "scan_maxpool3x3s2_forward": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "scan_upsample2x_add": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "scan_downsample2x_sum": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "scan_add_relu": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "scan_dbscan_prepare": (ctypes.c_int, [c_vp, c_i64, c_i32, c_f32, c_i32, c_vp, c_vp, c_vp]),
    "scan_dbscan_bfs_step": (ctypes.c_int, [c_i64, c_vp, c_i32, c_vp, c_vp]),
    "scan_dbscan_finish": (ctypes.c_int, [c_i64, c_vp, c_vp, c_vp]),
    "scan_sgd_momentum": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_i32, c_vp]),
    "scan_sgd_momentum_multi": (ctypes.c_int, [ctypes.POINTER(SgdSegment), c_i32, c_f32, c_vp]),
    "scan_weight_split_job_blocks": (c_i64, [c_i32, c_i32, c_i32, c_i32, c_i32]),
    "scan_weight_split_batched": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i64, c_vp]),
    "scan_cka_stack_weights": (ctypes.c_int, [ctypes.POINTER(CkaBranch), c_i32, c_i32, c_i32, c_i64, c_i64, c_i64, c_i64,
                                              c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "scan_cka_unstack_grads": (ctypes.c_int, [ctypes.POINTER(CkaBranch), c_i32, c_i32, c_i32, c_i64, c_i64, c_i64, c_i64,
                                              c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "scan_gconv3x3_to1_ws_floats": (c_i64, [_PD, c_i32, c_i32]),
    "scan_gconv3x3_to1_forward": (ctypes.c_int, [c_vp, _PD, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "scan_gconv3x3_to1_dgrad": (ctypes.c_int, [c_vp, c_i32, _PD, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "scan_gconv3x3_to1_wgrad": (ctypes.c_int, [c_vp, c_vp, c_i32, _PD, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp]),
    "scan_gconv3x3_to1_backward": (ctypes.c_int, [c_vp, c_vp, c_i32, _PD, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp,
                                                  c_vp]),
    "scan_gconv3x3_to1_forward_bits": (ctypes.c_int, [c_vp, _PD, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "scan_gconv3x3_to1_backward_bits": (ctypes.c_int, [c_vp, c_vp, c_i32, _PD, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32,
                                                       c_vp, c_vp]),
    "scan_resize_bilinear_u8": (ctypes.c_int, [c_vp, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp,
                                               c_i32, c_vp]),
    "scan_normalize_image_u8": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, ctypes.POINTER(c_f32),
                                               ctypes.POINTER(c_f32), c_vp, c_i32, c_i32, c_i32, c_vp]),
}

_lib = None


def lib():
    """Load libscan_hip.so; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "scan_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback." % LIB_PATH)
        # PyTorch-ROCm ships its own libamdhip64; load it first so this library binds to the SAME HIP runtime
        # (two runtimes in one process do not see each other's device context)
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        # SCAN_TUNE="key=value,key=value": launch-selection knobs for A/B measurements (scan_tune, include/scan_hip.h)
        for kv in filter(None, os.environ.get("SCAN_TUNE", "").split(",")):
            key, _, val = kv.partition("=")
            if L.scan_tune(key.strip().encode(), int(val)) == TUNE_UNKNOWN:
                raise RuntimeError("SCAN_TUNE: unknown key %r" % key)
        _lib = L
    return _lib


def tune_state():
    """{knob: (value, default)} of every scan_tune knob of the loaded library"""
    L, out, i = lib(), {}, 0
    while True:
        k = L.scan_tune_key(i)
        if k is None:
            return out
        out[k.decode()] = (L.scan_tune_get(k), L.scan_tune_default(k))
        i += 1


def lib_identity():
    """what a measurement was taken with: the library file (basename, sha1 of its bytes, whether it is the product library
    and not an alternative build selected by SCAN_HIP_LIB) and every scan_tune knob that is off its default"""
    import hashlib
    with open(LIB_PATH, "rb") as f:
        sha = hashlib.sha1(f.read()).hexdigest()
    product = os.path.realpath(LIB_PATH) == os.path.realpath(os.path.join(_HERE, "libscan_hip.so"))
    return {"lib": os.path.basename(LIB_PATH), "lib_sha1": sha, "product_library": product,
            "scan_tune_non_default": {k: v for k, (v, d) in tune_state().items() if v != d}}


def call(name, *args):
    """Call a status-returning entry point; raise RuntimeError on failure."""
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (name, rc, lib().scan_last_error().decode()))


def query(name, *args):
    return getattr(lib(), name)(*args)
