"""ctypes loader for oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see scan_oracle.c header).  numpy in, numpy out.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build():
    src = os.path.join(_HERE, "scan_oracle.c")
    if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        f32p = ctypes.POINTER(ctypes.c_float)
        i32p = ctypes.POINTER(ctypes.c_int32)
        i64p = ctypes.POINTER(ctypes.c_int64)
        L.oracle_nms.restype = ctypes.c_int64
        L.oracle_nms.argtypes = [f32p, f32p, ctypes.c_int64, ctypes.c_float, i64p]
        L.oracle_ml_nms.restype = ctypes.c_int64
        L.oracle_ml_nms.argtypes = [f32p, f32p, f32p, ctypes.c_int64, ctypes.c_float, i64p]
        L.oracle_sigmoid_focal_fwd.restype = None
        L.oracle_sigmoid_focal_fwd.argtypes = [f32p, i32p, ctypes.c_int64, ctypes.c_int32,
                                               ctypes.c_float, ctypes.c_float, f32p]
        L.oracle_sigmoid_focal_bwd.restype = None
        L.oracle_sigmoid_focal_bwd.argtypes = [f32p, i32p, f32p, ctypes.c_int64, ctypes.c_int32,
                                               ctypes.c_float, ctypes.c_float, f32p]
        L.oracle_iou_loss.restype = ctypes.c_double
        L.oracle_iou_loss.argtypes = [f32p, f32p, f32p, ctypes.c_int64, f32p]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def nms(dets, scores, thr):
    dets = _f32(dets).reshape(-1, 4)
    scores = _f32(scores).reshape(-1)
    n = dets.shape[0]
    keep = np.empty(max(n, 1), dtype=np.int64)
    k = lib().oracle_nms(_p(dets, ctypes.c_float), _p(scores, ctypes.c_float), n,
                         float(thr), _p(keep, ctypes.c_int64))
    return keep[:k].copy()


def ml_nms(dets, scores, labels, thr):
    dets = _f32(dets).reshape(-1, 4)
    scores = _f32(scores).reshape(-1)
    labels = _f32(labels).reshape(-1)
    n = dets.shape[0]
    keep = np.empty(max(n, 1), dtype=np.int64)
    k = lib().oracle_ml_nms(_p(dets, ctypes.c_float), _p(scores, ctypes.c_float),
                            _p(labels, ctypes.c_float), n, float(thr), _p(keep, ctypes.c_int64))
    return keep[:k].copy()


def sigmoid_focal_fwd(logits, targets, gamma, alpha):
    logits = _f32(logits)
    M, C = logits.shape
    targets = np.ascontiguousarray(targets, dtype=np.int32)
    out = np.empty_like(logits)
    lib().oracle_sigmoid_focal_fwd(_p(logits, ctypes.c_float), _p(targets, ctypes.c_int32), M, C,
                                   float(gamma), float(alpha), _p(out, ctypes.c_float))
    return out


def sigmoid_focal_bwd(logits, targets, d_losses, gamma, alpha):
    logits = _f32(logits)
    M, C = logits.shape
    targets = np.ascontiguousarray(targets, dtype=np.int32)
    d_losses = _f32(d_losses)
    out = np.empty_like(logits)
    lib().oracle_sigmoid_focal_bwd(_p(logits, ctypes.c_float), _p(targets, ctypes.c_int32),
                                   _p(d_losses, ctypes.c_float), M, C, float(gamma), float(alpha),
                                   _p(out, ctypes.c_float))
    return out


def iou_loss(pred, target, weight=None):
    pred = _f32(pred).reshape(-1, 4)
    target = _f32(target).reshape(-1, 4)
    P = pred.shape[0]
    losses = np.empty(max(P, 1), dtype=np.float32)
    wp = None
    if weight is not None:
        weight = _f32(weight).reshape(-1)
        wp = _p(weight, ctypes.c_float)
    v = lib().oracle_iou_loss(_p(pred, ctypes.c_float), _p(target, ctypes.c_float), wp, P,
                              _p(losses, ctypes.c_float))
    return float(v), losses[:P].copy()
