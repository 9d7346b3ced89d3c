"""The SCAN detector in the REFERENCE'S call shape on the drop-in operator surface.

BASELINE.json north_star: the kernels sit "behind the existing fcos_core.layers / fcos_core.modeling operator surface so
rpn/fcos/{condgraph,fcos,loss}.py call them unchanged".  ``scan_amd.engine`` does not run in that shape -- it keeps the whole
pyramid as one row matrix and launches once per layer for all five levels.  This module is the OTHER side of that claim: the
same sub-models (same parameters, same state_dict keys) driven the way the reference's module files drive theirs --

  * NCHW tensors, one module call per pyramid level (rpn/fcos/fcos.py:66-114 ``for l, feature in enumerate(x)``,
    condgraph.py:86-119, discriminator/fcos_head_discriminator_con.py:92-126 ``for c in range(self.num_classes)``),
  * ``nn.Sequential`` towers of Conv2d / GroupNorm / ReLU called as modules, torch's own ReLU / max-pool / interpolate / cat /
    BCE between them (backbone/mmdetection/vgg.py:154-169, backbone/fpn.py:67-91),
  * the predictions flattened level by level and concatenated in front of every loss (rpn/fcos/loss.py:191-202),

with ``torch.nn.Conv2d`` / ``nn.GroupNorm`` / ``F.conv2d`` swapped for ``scan_amd.layers.Conv2d`` / ``GroupNorm`` /
``dynamic_conv_softmax`` (the C++ autograd operators of scan_amd/ext/scan_ops/_ops when built) and the loss layers for
``scan_amd.layers.SigmoidFocalLoss`` / ``IOULoss`` / ``FocalLoss`` / ``GradientReversal``.  What it is for: ``bench.py --surface
layers`` times this graph beside the engine's (ms/step, launches/step, the per-operator table) so INTEGRATION.md can state
what a reference checkout that swaps its imports gets, and tests/test_gpu_model.py::test_surface_step_matches_engine_and_reference
holds its losses to the reference's fixture.  Host-side logic that is layout-free (ground-truth plan, graph tier, paradigm
update, kernel generator, loss evaluators) is shared with the engine: it consumes the flattened rows either way.
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import layers as L
from . import ops
from .modeling import fcos as fcos_mod
from .modeling.backbone import VGG_STAGES

LEVELS = ("P3", "P4", "P5", "P6", "P7")


def adopt(model):
    """Re-class every nn.Conv2d / nn.GroupNorm of the sub-models as scan_amd.layers.Conv2d / GroupNorm -- what swapping the
    import in a reference module file does.  Parameters, names and state_dict are untouched (the layers subclass torch's)."""
    for m in model.values():
        for sub in m.modules():
            if type(sub) is nn.Conv2d:
                sub.__class__ = L.Conv2d
            elif type(sub) is nn.GroupNorm:
                sub.__class__ = L.GroupNorm
                sub.fuse_relu = False
    return model


def _flatten(levels):
    """per-level [N, C, H, W] -> rows [sum N*H*W, C] in the order the reference concatenates in (level, image, y, x)."""
    return torch.cat([t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]) for t in levels], 0)


def backbone_forward(bb, images):
    """VGG16 body as mmcv's VGG.forward walks it (vgg.py:154-169: features one by one, the output of each stage's pooling
    kept) + FPN top-down (fpn.py:67-91) + LastLevelP6P7 on P5 (fpn.py:118-130)."""
    x = images.contiguous(memory_format=torch.channels_last)
    outs, feats = [], bb.body.features
    stage_end = {VGG_STAGES[s][-1] + 2: s for s in range(5)}  # index of the MaxPool2d closing each stage
    for i, layer in enumerate(feats):
        x = layer(x)
        if i in stage_end:
            outs.append(x)
    c3, c4, c5 = outs[2], outs[3], outs[4]
    fpn = bb.fpn
    last_inner = fpn.fpn_inner5(c5)
    results = [fpn.fpn_layer5(last_inner)]
    for feature, lvl in ((c4, 4), (c3, 3)):
        top_down = F.interpolate(last_inner, scale_factor=2, mode="nearest")
        last_inner = getattr(fpn, "fpn_inner%d" % lvl)(feature) + top_down
        results.insert(0, getattr(fpn, "fpn_layer%d" % lvl)(last_inner))
    p6 = fpn.top_blocks.p6(results[-1])
    p7 = fpn.top_blocks.p7(F.relu(p6))
    return results + [p6, p7]


def middle_head_forward(mh, feats_in, targets, shape, train_source=True):
    """GRAPHModule._forward_train_source / _forward_train_target(forward_target=False) (condgraph.py:321-384,500-534) per
    level: head_in tower, node sampling + graph tier + paradigm update on the flattened source rows, the conditioned kernels,
    dynamic conv + softmax per level, head_out on cat(features, act maps).
    -> (out features per level, act maps per level, node_loss or None, act_loss or None)"""
    feats = [mh.head_in.middle_tower(f) for f in feats_in]
    node_loss = act_loss = None
    if train_source:
        rows = _flatten(feats)
        plan = fcos_mod.target_plan(shape, targets, rows.device)
        node_loss, proto_batch = mh._forward_gcns(rows[plan.node_index], plan.node_labels)
        mh.update_prototype_nx1_rnn(proto_batch)
    kernels = mh.get_conded_weight()
    logits, maps, outs = [], [], []
    for f in feats:
        lg, pb = L.dynamic_conv_softmax(f, kernels)
        logits.append(lg)
        maps.append(pb)
        outs.append(mh.head_out.middle_tower(torch.cat([f, pb], 1)))
    if train_source:
        act_loss = mh.lamda2 * mh.act_loss_func(_flatten(logits), plan.labels.long())
    return outs, maps, node_loss, act_loss


def fcos_forward(fcos, feats, targets, shape):
    """FCOSHead.forward per level (fcos.py:66-114: both towers, cls_logits, exp(scale_l * bbox_pred), centerness on the
    regression tower) + FCOSLossComputation on the flattened, concatenated predictions (loss.py:191-230)."""
    head = fcos.head
    cls, reg, ctr = [], [], []
    for l, f in enumerate(feats):
        ct = head.cls_tower(f)
        bt = head.bbox_tower(f)
        cls.append(head.cls_logits(ct))
        ctr.append(head.centerness(bt))
        reg.append(torch.exp(head.scales[l](head.bbox_pred(bt))))
    lc, lr, lctr = fcos.loss_evaluator(shape, _flatten(cls), _flatten(reg), _flatten(ctr).reshape(-1), targets)
    return {"loss_cls": lc, "loss_reg": lr, "loss_centerness": lctr}


def discriminator_forward(dis, feature, act_maps, target):
    """FCOSDiscriminator_con.forward (fcos_head_discriminator_con.py:92-126) in the reference's own shape: GRL on both
    inputs, the tower, then PER foreground class cat(x, act[c + 1]) -> Conv2d(257, 128) -> ReLU -> Conv2d(128, 1) and the
    act-map-weighted BCE normalised by the map's sum, averaged over the classes (one class: plain mean BCE)."""
    feature = dis.grad_reverse(feature)
    act_maps = dis.grad_reverse(act_maps)
    x = dis.dis_tower(feature)
    loss = 0
    for c in range(dis.num_classes):
        a = act_maps[:, c + 1:c + 2]
        logit = getattr(dis, "classifier_cls_%d" % c)(torch.cat([x, a], 1))
        tgt = torch.full_like(logit, float(target))
        if dis.num_classes > 1:
            w = a.detach()
            lc = F.binary_cross_entropy_with_logits(logit, tgt, weight=w, reduction="sum") / w.sum()
        else:
            lc = F.binary_cross_entropy_with_logits(logit, tgt)
        loss = loss + lc / dis.num_classes
    return loss


class SurfaceTrainer:
    """One DA iteration (engine/trainer.py:266-424: source pass + its discriminators, target pass + its discriminators,
    every sub-model stepped once) over the graph above.  Gradient buffers, optimizer and schedule are the engine's
    (``trainer``: an engine.Trainer on the same model -- parameters are views of its flat buffers, autograd accumulates into
    its arena); only the forward / backward graph differs."""

    def __init__(self, trainer):
        self.trainer = trainer
        self.model = adopt(trainer.model)

    def _pass(self, images, targets, domain):
        model, lam = self.model, self.trainer.con_dis_lambda
        n = images.shape[0]
        feats_in = backbone_forward(model["backbone"], images)
        shape = ops.PyramidShape(n, [tuple(f.shape[-2:]) for f in feats_in])
        src = domain == "source"
        if src:
            fcos_mod.target_plan(shape, targets, images.device)
        feats, maps, node_loss, act_loss = middle_head_forward(model["middle_head"], feats_in, targets, shape, train_source=src)
        losses = {}
        if src:
            losses.update(node_loss_gs=node_loss, act_loss_gs=act_loss)
            losses.update({k + "_gs": v for k, v in fcos_forward(model["fcos"], feats, targets, shape).items()})
        else:
            losses["zero_gt"] = feats[0].new_zeros(())
        tag, label = ("ds", 1.0) if src else ("dt", 0.0)
        for lvl in ("P7", "P6", "P5", "P4", "P3"):
            i = LEVELS.index(lvl)
            losses["loss_adv_%s_CON_%s" % (lvl, tag)] = lam * discriminator_forward(model["dis_%s_CON" % lvl], feats[i], maps[i], label)
        return losses

    def step(self, images_s, targets_s, images_t):
        tr = self.trainer
        for m in self.model.values():
            m.train()
        fcos_mod.reset_target_plan()
        tr.grad_arena.zero_()
        out = self._pass(images_s, targets_s, "source")
        sum(out.values()).backward()
        lt = self._pass(images_t, None, "target")
        sum(v for k, v in lt.items() if k != "zero_gt").backward()
        out.update(lt)
        tr._optimizer_step()
        return out
