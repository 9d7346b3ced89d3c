"""Drop-in for the reference's pybind module ``fcos_core._C``
(reference fcos_core/csrc/vision.cpp:8-17): same names, argument meaning and
error behaviour, backed by libscan_hip.so through the C ABI.

    nms(dets[n,4], scores[n], thr) -> int64[k]
    ml_nms(dets[n,4], scores[n], labels[n] float, thr) -> int64[k]
    sigmoid_focalloss_forward(logits[M,C], targets[M] int32, num_classes, gamma, alpha) -> losses[M,C]
    sigmoid_focalloss_backward(logits, targets, d_losses, num_classes, gamma, alpha) -> d_logits[M,C]

roi_align_* / roi_pool_* belong to the two-stage heads no SCAN config uses
 and raise.
"""
import torch

from . import ops
from ._lib import call
from .ops import _ptr, _stream


def _require_gpu(t, who):
    if not t.is_cuda:
        # the reference raises "Not implemented on the CPU" (csrc/SigmoidFocalLoss.h:23, csrc/ml_nms.h:26)
        raise RuntimeError("%s: not implemented on the CPU" % who)


def nms(dets, scores, threshold):
    """Greedy NMS; IoU >= threshold suppresses (the reference CPU rule, csrc/cpu/nms_cpu.cpp:60,
    which is what the oracle pins).  Empty input returns an empty CPU tensor (csrc/nms.h:17-18)."""
    if dets.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device="cpu")
    _require_gpu(dets, "nms")
    return ops.nms(dets, scores, threshold, rule_ge=True)


def ml_nms(dets, scores, labels, threshold):
    if dets.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device="cpu")
    _require_gpu(dets, "ml_nms")
    return ops.ml_nms(dets, scores, labels, threshold)


def sigmoid_focalloss_forward(logits, targets, num_classes, gamma, alpha):
    _require_gpu(logits, "sigmoid_focalloss_forward")
    if logits.dim() != 2:
        raise RuntimeError("logits should be NxClass")  # SigmoidFocalLoss_cuda.cu:112
    if logits.shape[1] != num_classes:
        raise RuntimeError("logits.size(1) should be num_classes")
    logits = logits.contiguous()
    targets = targets.contiguous()
    losses = torch.empty_like(logits)
    call("scan_sigmoid_focal_loss_forward", _ptr(logits), _ptr(targets), logits.shape[0], num_classes, float(gamma),
         float(alpha), _ptr(losses), None, _stream())
    return losses


def sigmoid_focalloss_backward(logits, targets, d_losses, num_classes, gamma, alpha):
    _require_gpu(logits, "sigmoid_focalloss_backward")
    if logits.dim() != 2 or logits.shape[1] != num_classes:
        raise RuntimeError("logits.size(1) should be num_classes")  # SigmoidFocalLoss_cuda.cu:158
    logits = logits.contiguous()
    targets = targets.contiguous()
    d_losses = d_losses.contiguous()
    d_logits = torch.empty_like(logits)
    call("scan_sigmoid_focal_loss_backward", _ptr(logits), _ptr(targets), _ptr(d_losses), 1.0, logits.shape[0],
         num_classes, float(gamma), float(alpha), _ptr(d_logits), _stream())
    return d_logits


def _two_stage(*a, **k):
    raise RuntimeError("roi_align/roi_pool are outside the SCAN hot path (RPN_ONLY configs); not built")


roi_align_forward = roi_align_backward = roi_pool_forward = roi_pool_backward = _two_stage
