"""Operator surface mirroring ``fcos_core.layers`` for the SCAN hot path
(reference fcos_core/layers/__init__.py:4-27): same class names, constructor
arguments and forward semantics, HIP underneath."""
import torch
from torch import nn

from .. import _C as _C_ctypes, ops

# ``_C``: the compiled pybind11 module a reference checkout imports as fcos_core._C (scan_amd/csrc/fcos_core_C.cpp, built by
# __graft_entry__.build() into scan_amd/ext/fcos_core/); scan_amd/_C.py is the same four functions bound through ctypes
# (kept for environments without a C++ toolchain -- both sit on the C ABI of libscan_hip.so, neither computes anything itself)
try:
    from ..ext.fcos_core import _C
    C_BACKEND = "compiled"
except ImportError:
    _C = _C_ctypes
    C_BACKEND = "ctypes"

nms = _C.nms
ml_nms = _C.ml_nms

# ``_ops``: the conv / GroupNorm / dynamic-conv operators with C++ autograd (scan_amd/csrc/scan_ops_ext.cpp ->
# scan_amd/ext/scan_ops/_ops<EXT_SUFFIX>): the NCHW drop-in modules below run forward and backward on it without entering
# the Python interpreter between kernel launches.  Without a C++ toolchain (or with SCAN_OPS_BACKEND=python) the same
# modules go through scan_amd.ops (Python autograd.Function -> ctypes): same kernels, bit-identical results.
import os as _os
try:
    if _os.environ.get("SCAN_OPS_BACKEND", "") == "python":
        raise ImportError("SCAN_OPS_BACKEND=python")
    from ..ext.scan_ops import _ops
    from .._lib import lib as _lib_handle
    # a module left over from an earlier build must not run against a newer library: both report the ABI they were built for
    if int(_ops.scan_abi_version()) != int(_lib_handle().scan_abi_version()):
        raise ImportError("scan_ops._ops was built for ABI %d, libscan_hip.so is ABI %d: rebuild (__graft_entry__.build())"
                          % (_ops.scan_abi_version(), _lib_handle().scan_abi_version()))
    OPS_BACKEND = "compiled"
    ops._EXT_INVALIDATE = _ops.invalidate_weight_cache  # ops.invalidate_weight_planes() then drops the C++ side's planes too
except ImportError as _e:
    if "ABI" in str(_e):
        import warnings as _warnings
        _warnings.warn(str(_e) + " -- scan_amd.layers uses its ctypes bindings")
    _ops = None
    OPS_BACKEND = "python"


class SigmoidFocalLoss(nn.Module):
    """reference layers/sigmoid_focal_loss.py:56-70: returns the SUM of the element losses."""

    def __init__(self, gamma, alpha):
        super().__init__()
        self.gamma = gamma
        self.alpha = alpha

    def forward(self, logits, targets):
        return ops.sigmoid_focal_loss_sum(logits, targets, self.gamma, self.alpha)

    def __repr__(self):
        return "%s(gamma=%s, alpha=%s)" % (self.__class__.__name__, self.gamma, self.alpha)


class IOULoss(nn.Module):
    """reference layers/iou_loss.py:5-36."""

    def forward(self, pred, target, weight=None):
        assert pred.numel() != 0
        return ops.iou_loss(pred, target, weight)


class FocalLoss(nn.Module):
    """The FocalLoss actually bound by the reference (layers/__init__.py:24 ->
    layers/sigmoid_focal_loss_wbg.py:7-64): softmax focal loss, alpha = 1, mean."""

    def __init__(self, class_num, alpha=None, gamma=2, size_average=True):
        super().__init__()
        assert alpha is None and size_average, "only the configuration SCAN uses is built"
        self.class_num = class_num
        self.gamma = gamma

    def forward(self, inputs, targets):
        return ops.softmax_focal_loss_mean(inputs, targets, self.gamma)


class Scale(nn.Module):
    """reference layers/scale.py:5-11."""

    def __init__(self, init_value=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.FloatTensor([init_value]))

    def forward(self, input):
        return input * self.scale


class GradientReversal(nn.Module):
    """reference modeling/discriminator/layer.py:27-33."""

    def __init__(self, lambda_=1):
        super().__init__()
        self.lambda_ = lambda_

    def forward(self, x):
        return ops.grad_reverse(x, self.lambda_)


class MultiHeadAttention(nn.Module):
    """Graph-node aggregation (reference layers/transformer.py:36-90).  Stays on the torch
    tier (north star: node aggregation is host-side glue): a handful of [n,256] matmuls.
    Quirks kept: heads are a plain .view(heads,-1,64) of the [1,n,256] tensor; scale is
    (64 // heads) ** -0.5; dropout on attention and output."""

    def __init__(self, model_dim=400, num_heads=4, dropout=0.0):
        super().__init__()
        self.dim_per_head = model_dim // num_heads
        self.num_heads = num_heads
        self.linear_k = nn.Linear(model_dim, self.dim_per_head * num_heads)
        self.linear_v = nn.Linear(model_dim, self.dim_per_head * num_heads)
        self.linear_q = nn.Linear(model_dim, self.dim_per_head * num_heads)
        self.linear_final = nn.Linear(model_dim, model_dim)
        self.dropout = nn.Dropout(dropout)
        self.attn_dropout = nn.Dropout(dropout)
        self.layer_norm = nn.LayerNorm(model_dim)

    def forward(self, key, value, query, attn_mask=None):
        residual = query
        d, h = self.dim_per_head, self.num_heads
        b = key.size(0)
        key = self.linear_k(key).view(b * h, -1, d)
        value = self.linear_v(value).view(b * h, -1, d)
        query = self.linear_q(query).view(b * h, -1, d)
        scale = (key.size(-1) // h) ** -0.5
        attention = self.attn_dropout(torch.softmax(torch.bmm(query, key.transpose(1, 2)) * scale, dim=2))
        context = torch.bmm(attention, value).view(b, -1, d * h)
        output = self.dropout(self.linear_final(context))
        return self.layer_norm(residual + output), attention


# ----------------------------------------------------------------------------- NCHW drop-in modules
# The reference's modules call torch.nn.Conv2d / nn.GroupNorm / F.conv2d on NCHW tensors (rpn/fcos/fcos.py:25-64,
# rpn/fcos/condgraph.py:86-106,619-629, discriminator/fcos_head_discriminator_con.py:20-87).  These classes keep those
# constructor signatures and the NCHW call convention, so such a module file runs on the HIP kernels by swapping the
# import.  A torch.channels_last NCHW tensor IS the kernels' pixel-major row matrix, so the adaptor
# is a view in both directions; any other layout is converted once on entry.
def _to_rows(x):
    """[N,C,H,W] -> (rows [N*H*W, Cs], PyramidShape, C): zero-copy for channels_last tensors with C % 4 == 0."""
    n, c, h, w = x.shape
    rows = x.permute(0, 2, 3, 1)
    if c % 4 != 0:
        rows = torch.nn.functional.pad(rows, (0, ops.pad4(c) - c))
    rows = rows.contiguous().view(n * h * w, -1)
    return rows, ops.PyramidShape(n, [(h, w)]), c


def _to_nchw(rows, shape, c):
    h, w = shape.sizes[0]
    return rows.view(shape.n_images, h, w, rows.shape[1])[..., :c].permute(0, 3, 1, 2)  # channels_last NCHW view


class Conv2d(nn.Conv2d):
    """torch.nn.Conv2d(in_channels, out_channels, kernel_size, stride=1, padding=0, ...) on the MFMA conv kernels:
    kernel 1 / 3 (5 / 7 through the generic kernel), stride 1 / 2, padding = kernel // 2, dilation 1, groups 1 -- the
    convolutions the SCAN modules build.  Input and output are NCHW (channels_last memory format)."""

    def forward(self, x):
        k, s = self.kernel_size[0], self.stride[0]
        if (self.kernel_size[0] != self.kernel_size[1] or self.stride[0] != self.stride[1] or self.groups != 1
                or tuple(self.dilation) != (1, 1) or tuple(self.padding) != (k // 2, k // 2) or s not in (1, 2)
                or self.padding_mode != "zeros"):
            raise RuntimeError("scan_amd.layers.Conv2d: only square kernels, stride 1 / 2, padding = k // 2, "
                               "dilation 1, groups 1 are built (what the SCAN modules use)")
        if _ops is not None and k in (1, 3) and ops.CONV_MODE == "bf16x6":
            return _ops.conv2d(x, self.weight, self.bias, s, False)
        rows, shape, _ = _to_rows(x)
        w = self.weight
        if not w.permute(0, 2, 3, 1).is_contiguous():  # kernels read [Cout][k*k][Cin]
            w = w.contiguous(memory_format=torch.channels_last)
        y = ops.conv2d(rows, w, self.bias, shape, k, s)
        return _to_nchw(y, shape.conv_out(k, s), self.out_channels)


class GroupNorm(nn.GroupNorm):
    """torch.nn.GroupNorm(32, 256) (+ the ReLU that follows it in every SCAN tower when relu=True) on NCHW tensors."""

    def __init__(self, num_groups, num_channels, eps=1e-5, affine=True, relu=False):
        super().__init__(num_groups, num_channels, eps, affine)
        self.fuse_relu = relu

    def forward(self, x):
        # the kernels are built for the one GroupNorm the SCAN modules use: 32 groups, affine, whole float4 channel groups
        if self.num_groups != 32 or not self.affine or self.num_channels % 32 != 0 or x.shape[1] != self.num_channels:
            raise RuntimeError("scan_amd.layers.GroupNorm: only GroupNorm(32, C) with affine=True and C %% 32 == 0 is built "
                               "(got num_groups=%d, num_channels=%d, affine=%s, input channels=%d)"
                               % (self.num_groups, self.num_channels, self.affine, x.shape[1]))
        if _ops is not None:
            return _ops.group_norm_relu(x, self.weight, self.bias, self.eps, self.fuse_relu)
        rows, shape, c = _to_rows(x)
        y = ops.groupnorm_relu(rows, self.weight, self.bias, shape, relu=self.fuse_relu, eps=self.eps)
        return _to_nchw(y, shape, c)


def conv3x3_gn_relu(x, conv, gn, relu=True):
    """The tower block of every SCAN head -- ``[nn.Conv2d(C, 256, 3, 1, 1), nn.GroupNorm(32, 256), nn.ReLU()]``
    (rpn/fcos/fcos.py:36-49, rpn/fcos/condgraph.py:99-105, discriminator/fcos_head_discriminator_con.py:20-34) -- as ONE
    operator: the GroupNorm statistics come out of the conv's epilogue, normalisation + ReLU are one pass.  ``conv`` / ``gn``:
    modules (or anything with .weight / .bias, gn.eps) holding the parameters; x and the result are NCHW."""
    if _ops is not None and ops.CONV_MODE == "bf16x6":
        return _ops.conv3x3_gn_relu(x, conv.weight, conv.bias, gn.weight, gn.bias, gn.eps, relu)
    rows, shape, _ = _to_rows(x)
    w = conv.weight
    if not w.permute(0, 2, 3, 1).is_contiguous():
        w = w.contiguous(memory_format=torch.channels_last)
    y = ops.conv2d(rows, w, conv.bias, shape, 3, 1, gn_sums=True)
    y = ops.groupnorm_relu(y, gn.weight, gn.bias, shape, relu=relu, eps=gn.eps)
    return _to_nchw(y, shape, conv.weight.shape[0])


def dynamic_conv_softmax(features, kernel_par):
    """GRAPHModule.dynamic_conv + softmax(dim=1) (reference condgraph.py:619-629, 344-346): features [N,256,H,W],
    kernel_par [K,256] -> (act-map logits [N,K,H,W], act maps [N,K,H,W])."""
    if _ops is not None:
        return tuple(_ops.dynamic_conv_softmax(features, kernel_par))
    rows, shape, _ = _to_rows(features)
    logits, probs = ops.dynconv_softmax(rows, kernel_par)
    h, w = shape.sizes[0]
    back = lambda t: t.view(shape.n_images, h, w, t.shape[1]).permute(0, 3, 1, 2)
    return back(logits), back(probs)
