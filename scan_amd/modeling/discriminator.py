"""Conditional Kernel-guided Alignment (CKA) discriminator on one pyramid level.

Mirrors the reference's FCOSDiscriminator_con (fcos_core/modeling/discriminator/
fcos_head_discriminator_con.py:12-126) for the shipped setting (fusion 'concat', GRL on both
inputs, use_bg False, num_classes > 1) with the reference's parameter names
(dis_tower.N, classifier_cls_C.{0,2}).

MI355X restructuring (same arithmetic, fewer and fatter launches): the reference runs, per
foreground class c, cat(x, act[c+1]) -> conv3x3 257->128 -> ReLU -> conv3x3 128->1.  All
classes read the same x, so the 8 first convs are ONE implicit GEMM with 8*128 output channels
over cat(x, act[1:]) (the per-class act channel enters through a block-diagonal weight
slice), and the 8 second convs are ONE skinny conv 1024 -> 8 with a block-diagonal weight.
The class-weighted BCE of all classes is one wavefront-reduction kernel.
"""
import torch
from torch import nn

from .. import ops
from ..layers import GradientReversal
from .backbone import conv_holder
from .fcos import make_tower, run_tower


class FCOSDiscriminator_con(nn.Module):
    def __init__(self, with_GA=False, fusion_cfg="concat", num_convs=4, in_channels=256, num_classes=9,
                 grad_reverse_lambda=0.02, grl_applied_domain="both", patch_stride=None, cfg=None):
        super().__init__()
        assert fusion_cfg == "concat" and grl_applied_domain == "both" and patch_stride is None and num_classes >= 2, \
            "only the configuration the SCAN yamls use is built"
        self.num_convs = num_convs
        self.in_channels = in_channels
        self.num_classes = num_classes - 1  # use_bg False
        self.dis_tower = make_tower(num_convs, in_channels)
        for c in range(self.num_classes):
            self.add_module("classifier_cls_%d" % c, nn.Sequential(
                conv_holder(in_channels + 1, 128, 3), nn.ReLU(), conv_holder(128, 1, 3)))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, std=0.01)
                nn.init.constant_(m.bias, 0)
        self.grad_reverse = GradientReversal(grad_reverse_lambda)

    def _stacked_weights_torch(self):
        """the stacked weights spelled out in torch ops (what ops.cka_stacked_weights computes in one launch); kept as
        the readable definition and for the test that compares the two"""
        Cf, C = self.num_classes, self.in_channels
        blocks = [getattr(self, "classifier_cls_%d" % c) for c in range(Cf)]
        w0 = torch.stack([b[0].weight for b in blocks], 0)  # [Cf,128,C+1,3,3]
        main = w0[:, :, :C].reshape(Cf * 128, C, 3, 3)
        extra = w0[:, :, C]  # [Cf,128,3,3]: the act-map input channel of each class
        eye = torch.eye(Cf, device=w0.device, dtype=w0.dtype)
        blk = (extra[:, :, None] * eye[:, None, :, None, None]).reshape(Cf * 128, Cf, 3, 3)
        w1 = torch.cat([main, blk], 1)  # [Cf*128, C+Cf, 3, 3]
        b1 = torch.cat([b[0].bias for b in blocks], 0)
        w2s = torch.stack([b[2].weight[0] for b in blocks], 0)  # [Cf,128,3,3]
        w2 = (w2s[:, None] * eye[:, :, None, None, None]).reshape(Cf, Cf * 128, 3, 3)
        b2 = torch.cat([b[2].bias for b in blocks], 0)
        return w1, b1, w2, b2

    def _logits(self, feature, act_maps, shape, grl_applied=False):
        Cf = self.num_classes
        if not grl_applied:  # the caller already put both inputs behind this module's GRL (ops.split_levels_grl)
            feature = self.grad_reverse(feature)
            act_maps = self.grad_reverse(act_maps)
        # cat([x, act[1:]], 1) (reference :104-118) without copying x: the tower's last GroupNorm writes into the first
        # 256 columns of the [M, pad4(256 + Cf)] class-branch input, only the Cf act columns are copied behind them, and
        # the backward hands the GroupNorm its column slice of the conv's data gradient in place
        cs1 = ops.pad4(self.in_channels + Cf)
        if ops.CAT_IN_PLACE and self.num_convs > 0:
            buf = feature.new_empty((feature.shape[0], cs1))
            x = run_tower(self.dis_tower, feature, shape, self.num_convs, out_buf=buf)
            xcat = ops.cat_into(x, act_maps[:, 1:], buf)
        else:
            x = run_tower(self.dis_tower, feature, shape, self.num_convs)
            xcat = torch.cat([x, act_maps[:, 1:]], 1)  # [M, 256 + Cf]; 264 is a multiple of 4
            if cs1 != xcat.shape[1]:
                xcat = torch.nn.functional.pad(xcat, (0, cs1 - xcat.shape[1]))
        blocks = [getattr(self, "classifier_cls_%d" % c) for c in range(Cf)]
        if ops.BATCHED:
            w1, b1, w2, b2 = ops.cka_stacked_weights([(b[0], b[2]) for b in blocks], self.in_channels, 128,
                                                     xcat.shape[1])
        else:
            w1, b1, w2, b2 = self._stacked_weights_torch()
        h = ops.conv2d(xcat, w1, b1, shape, 3, 1, relu="deferred")  # its only consumer masks dx by (h > 0)
        if ops.GROUPED_CLS and Cf in (1, 2, 4, 8):
            # one output channel per class from that class's 128 hidden channels: an HBM-bound grouped kernel instead of
            # a dense conv over the block-diagonal weight (64x the useful multiply-adds)
            return ops.gconv3x3_to1(h, w2, b2, shape, Cf, mask_dx=True)[:, :Cf], act_maps
        return ops.conv2d(h, w2, b2, shape, 3, 1, mask_dx=True)[:, :Cf], act_maps

    def _loss(self, logits, act_maps, target):
        Cf = self.num_classes
        if Cf == 1:
            # single foreground class (Sim10k / KITTI): plain mean BCE, no act-map weight (reference :122-123)
            logits = logits.reshape(-1)
            return ops.bce_with_logits_mean(logits, torch.full_like(logits, float(target)))
        return ops.cka_bce(logits, act_maps.detach(), float(target), Cf)

    def forward(self, feature, target, act_maps=None, domain="source", shape=None):
        """feature [M_l,256], act_maps [M_l,K] rows of ONE level; shape = that level's PyramidShape."""
        assert target in (0, 1, 0.1, 0.9) and domain in ("source", "target")
        logits, act_maps = self._logits(feature, act_maps, shape)
        return self._loss(logits, act_maps, target)

    def forward_pair(self, feature, act_maps, shape, n_src, grl_applied=False):
        """source frames [0, n_src) and target frames [n_src, N) of one level in ONE pass through the tower and the
        class branches; the two domain losses (labels 1.0 / 0.0, each with its own act-map normaliser) are taken on
        the two halves of the rows.  Same values and gradients as forward(.., 1.0, 'source') + forward(.., 0.0,
        'target') (reference trainer.py:314-333, 373-376)."""
        logits, act_maps = self._logits(feature, act_maps, shape, grl_applied)
        (h, w) = shape.sizes[0]
        m = n_src * h * w
        if self.num_classes > 1 and 0 < m < logits.shape[0]:
            # both domain losses as one node: its backward writes the two halves of ONE gradient buffer (ops.cka_bce_pair)
            return ops.cka_bce_pair(logits, act_maps.detach(), m, self.num_classes)
        ls, lt = ops.split_rows2(logits, m)  # one gradient buffer in the backward instead of two zero-filled ones + an add
        return self._loss(ls, act_maps[:m], 1.0), self._loss(lt, act_maps[m:], 0.0)
