"""Operator surface mirroring ``fcos_core.layers`` for the SCAN hot path
(reference fcos_core/layers/__init__.py:4-27): same class names, constructor
arguments and forward semantics, HIP underneath."""
import torch
from torch import nn

from .. import _C, ops

nms = _C.nms
ml_nms = _C.ml_nms


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
