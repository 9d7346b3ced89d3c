"""Drop-in for the reference's pybind module ``fcos_core._C``
(reference fcos_core/csrc/vision.cpp:8-17): same names, argument meaning and
error behaviour, backed by libscan_hip.so through the C ABI.

    nms(dets[n,4], scores[n], thr[, cuda_rule]) -> int64[k]     (CPU tensors: the host loop of nms_host below)
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


def nms_host(dets, scores, threshold, rule_ge=True):
    """The reference's CPU dispatch (csrc/nms.h:26, csrc/cpu/nms_cpu.cpp:5-65) for host tensors: candidates by
    descending score (ties: lower index first), a kept box suppresses every later box whose IoU -- areas with the +1
    pixel rule, in the tensors' own dtype -- reaches the threshold; kept ORIGINAL indices ascending.  One vectorised
    row of IoUs per kept box (same operations in the same order as the scalar loop, so the same keep list)."""
    if scores.is_cuda:
        raise RuntimeError("scores must be a CPU tensor")           # nms_cpu.cpp:10
    if dets.dtype != scores.dtype:
        raise RuntimeError("dets should have the same type as scores")  # nms_cpu.cpp:11
    if dets.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("nms: float or double boxes")
    n = dets.shape[0]
    by_score = torch.sort(scores, descending=True, stable=True).indices
    b = dets[by_score].contiguous()
    area = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    dead = torch.zeros(n, dtype=torch.bool)
    zero = torch.zeros((), dtype=dets.dtype)
    thr = float(torch.tensor(threshold, dtype=torch.float32))  # the reference's `const float threshold`
    for r in range(n):
        if dead[r]:
            continue
        rest = b[r + 1:]
        w = torch.maximum(zero, torch.minimum(b[r, 2], rest[:, 2]) - torch.maximum(b[r, 0], rest[:, 0]) + 1)
        h = torch.maximum(zero, torch.minimum(b[r, 3], rest[:, 3]) - torch.maximum(b[r, 1], rest[:, 1]) + 1)
        inter = w * h
        iou = inter / (area[r] + area[r + 1:] - inter)
        dead[r + 1:] |= (iou >= thr) if rule_ge else (iou > thr)
    return torch.sort(by_score[~dead]).values


def _cuda_rule_env():
    import os
    return os.environ.get("SCAN_NMS_RULE", "") in ("gt", "cuda")


def nms(dets, scores, threshold, cuda_rule=False):
    """Greedy NMS.  Default: IoU >= threshold suppresses (the reference CPU rule, csrc/cpu/nms_cpu.cpp:60, which its
    tests/test_nms.py pins); ``cuda_rule=True`` or SCAN_NMS_RULE=gt: IoU > threshold (csrc/cuda/nms.cu:60).  GPU tensors
    run on libscan_hip.so, CPU tensors on ``nms_host`` (csrc/nms.h:10-30 dispatches the same way).  Empty input returns
    an empty CPU tensor (csrc/nms.h:17-18)."""
    if dets.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device="cpu")
    rule_ge = not (cuda_rule or _cuda_rule_env())
    if not dets.is_cuda:
        return nms_host(dets, scores, threshold, rule_ge)
    return ops.nms(dets, scores, threshold, rule_ge=rule_ge)


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
