"""Authoring-container-only harness that imports the reference's Python hot path
on CPU -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

It exists to (1) validate oracle/scan_ref.py (our torch-CPU restatement) and
oracle/scan_oracle.c against the real reference and (2) generate the golden
vectors under tests/golden/ (see oracle/make_golden.py).  It reads
/root/reference, which does not exist on the GPU box; nothing in tests -m gpu,
smoke() or bench.py imports this file.

The reference is not runnable unmodified on CPU / modern torch:
the stubs and monkeypatches below are the minimum needed and change no
arithmetic on the path.
"""
import ast
import os
import re
import sys
import types

import torch
import yaml

REF = os.environ.get("SCAN_REFERENCE", "/root/reference")
_READY = False


class CfgNode(dict):
    """Minimal yacs.config.CfgNode stand-in (attribute access, merge, clone)."""

    def __init__(self, init=None):
        super().__init__()
        if init:
            for k, v in init.items():
                self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        out = CfgNode()
        for k, v in self.items():
            out[k] = v.clone() if isinstance(v, CfgNode) else v
        return out

    def freeze(self):
        pass

    def defrost(self):
        pass

    @staticmethod
    def _coerce(new, old):
        if isinstance(new, str):
            try:
                new = ast.literal_eval(new)
            except Exception:
                pass
        if isinstance(old, tuple) and isinstance(new, list):
            new = tuple(new)
        if isinstance(old, list) and isinstance(new, tuple):
            new = list(new)
        return new

    def _merge(self, d):
        for k, v in d.items():
            if isinstance(v, dict):
                if k not in self:
                    self[k] = CfgNode()
                self[k]._merge(v)
            else:
                self[k] = self._coerce(v, self.get(k))

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(yaml.safe_load(f))

    def merge_from_list(self, lst):
        for key, val in zip(lst[0::2], lst[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = self._coerce(val, node.get(parts[-1]))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _RefC:
    """fcos_core._C stand-in.  nms = OUR C oracle (validated against the
    reference's tests/test_nms.py known answers by tests/test_oracle.py); the
    reference's CPU build raises for the rest (csrc/ml_nms.h:26,
    csrc/SigmoidFocalLoss.h:23), so do we."""

    @staticmethod
    def nms(dets, scores, thr):
        from oracle import coracle
        if dets.numel() == 0:
            return torch.empty((0,), dtype=torch.int64)
        return torch.from_numpy(coracle.nms(dets.detach().numpy(), scores.detach().numpy(), thr))

    @staticmethod
    def ml_nms(*a, **k):
        raise RuntimeError("CPU version not implemented")

    @staticmethod
    def sigmoid_focalloss_forward(*a, **k):
        raise RuntimeError("Not implemented on the CPU")

    sigmoid_focalloss_backward = sigmoid_focalloss_forward
    roi_align_forward = roi_align_backward = sigmoid_focalloss_forward
    roi_pool_forward = roi_pool_backward = sigmoid_focalloss_forward


def setup():
    """Install stubs + monkeypatches; idempotent."""
    global _READY
    if _READY:
        return
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.insert(0, repo)

    yacs = _mod("yacs")
    yacs.config = _mod("yacs.config", CfgNode=CfgNode)
    _mod("ipdb")
    if "cv2" not in sys.modules:  # imported by structures/segmentation_mask.py on the way to fcos_core.data.transforms; unused
        _mod("cv2")
    six = _mod("torch._six", PY3=True, string_classes=(str,), int_classes=(int,))
    torch._six = six

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    pc = _mod("pycocotools")
    pc.coco = _mod("pycocotools.coco", COCO=_Dummy)
    pc.mask = _mod("pycocotools.mask")
    pc.cocoeval = _mod("pycocotools.cocoeval", COCOeval=_Dummy)
    tv = _mod("torchvision")
    tv.transforms = _mod("torchvision.transforms")
    # torchvision is absent from this image.  The reference's transforms (data/transforms/transforms.py:27-90) call
    # four of its functional ops on PIL images; their PIL code path (torchvision/transforms/functional.py +
    # _functional_pil.py, unchanged across 0.2 ... 0.20) is restated here on the REAL PIL that is installed:
    #   resize(img, (h, w))  -> img.resize((w, h), BILINEAR)       (the default interpolation)
    #   hflip(img)           -> img.transpose(FLIP_LEFT_RIGHT)
    #   to_tensor(img)       -> uint8 HWC -> float CHW, .div(255)
    #   normalize(t, m, s)   -> (t - m[:, None, None]) / s[:, None, None]
    def _tv_resize(img, size, interpolation=None):
        from PIL import Image
        return img.resize((int(size[1]), int(size[0])), Image.BILINEAR)

    def _tv_hflip(img):
        from PIL import Image
        return img.transpose(Image.FLIP_LEFT_RIGHT)

    def _tv_to_tensor(img):
        import numpy as _np
        a = torch.from_numpy(_np.asarray(img).copy())
        return a.permute(2, 0, 1).contiguous().float().div(255)

    def _tv_normalize(t, mean, std):
        m = torch.as_tensor(mean, dtype=t.dtype)[:, None, None]
        sd = torch.as_tensor(std, dtype=t.dtype)[:, None, None]
        return (t - m) / sd

    tv.transforms.functional = _mod("torchvision.transforms.functional", resize=_tv_resize, hflip=_tv_hflip,
                                    to_tensor=_tv_to_tensor, normalize=_tv_normalize)
    tv.datasets = _mod("torchvision.datasets")
    tv.datasets.coco = _mod("torchvision.datasets.coco", CocoDetection=_Dummy)
    tv.datasets.CocoDetection = _Dummy
    for name in ("matplotlib", "matplotlib.pyplot"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _mod(name)
    if not hasattr(torch.hub, "_download_url_to_file"):
        torch.hub._download_url_to_file = lambda *a, **k: None

    # hard-coded .cuda()/.to('cuda') in condgraph.py:170,198-200,234,237 and
    # loss.py:80,82,92,162-164 -> identity on this CPU-only container
    _orig_to = torch.nn.Module.to

    def _to(self, *args, **kwargs):
        args = tuple(a for a in args if not (a == "cuda" or (isinstance(a, torch.device) and a.type == "cuda")))
        if kwargs.get("device") in ("cuda", torch.device("cuda")):
            kwargs.pop("device")
        if not args and not kwargs:
            return self
        return _orig_to(self, *args, **kwargs)

    torch.nn.Module.to = _to
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.cuda = lambda self, *a, **k: self

    import fcos_core  # noqa: F401  (package __init__ is empty)
    sys.modules["fcos_core._C"] = _RefC
    fcos_core._C = _RefC
    _READY = True


def make_cfg(extra=(), yaml_name="scan_vgg16_cityscapace_to_foggy.yaml"):
    setup()
    from fcos_core.config import cfg as _cfg
    cfg = _cfg.clone()
    path = os.path.join(REF, "configs/scan", yaml_name)
    text = open(path).read()
    # scan_vgg16_sim10k_to_cityscapes.yaml:5 carries a stray extra space before WEIGHT (not valid YAML as shipped):
    # parse an in-memory copy with that one line re-indented
    fixed = re.sub(r"(?m)^   WEIGHT:", "  WEIGHT:", text)
    if fixed != text:
        import tempfile
        with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
            f.write(fixed)
            path = f.name
    cfg.merge_from_file(path)
    cfg.merge_from_list(["MODEL.DEVICE", "cpu", "MODEL.WEIGHT", ""] + list(extra))
    return cfg


class _CosEmb(torch.nn.Module):
    """condgraph.py:479-480 passes a [1,K^2] target; modern torch wants [1].
    Old semantics broadcast to the same mean value."""

    def forward(self, a, b, t):
        return torch.nn.functional.cosine_embedding_loss(a, b, t[:, 0], margin=0.0)


def build_models(cfg, dropout=0.0):
    """backbone, middle_head, fcos, 5 CKA discriminators of the reference."""
    setup()
    from fcos_core.modeling.backbone import build_backbone
    from fcos_core.modeling.rpn.rpn import build_rpn, build_middle_head
    from fcos_core.modeling.discriminator.fcos_head_discriminator_con import FCOSDiscriminator_con

    model = {}
    model["backbone"] = build_backbone(cfg)
    model["middle_head"] = build_middle_head(cfg, 256)
    model["fcos"] = build_rpn(cfg, 256)
    for lvl in ("P7", "P6", "P5", "P4", "P3"):
        model["dis_%s_CON" % lvl] = FCOSDiscriminator_con(
            with_GA=cfg.MODEL.ADV.CON_WITH_GA, fusion_cfg=cfg.MODEL.ADV.CON_FUSUIN_CFG,
            num_convs=4, grad_reverse_lambda=0.02, grl_applied_domain=cfg.MODEL.ADV.GRL_APPLIED_DOMAIN,
            num_classes=cfg.MODEL.FCOS.NUM_CLASSES, cfg=cfg)
    lf = model["fcos"].loss_evaluator.cls_loss_func
    # sigmoid_focal_loss_cpu does gamma[0] (layers/sigmoid_focal_loss.py:43-44)
    lf.gamma = [lf.gamma]
    lf.alpha = [lf.alpha]
    mh = model["middle_head"]
    if hasattr(mh, "transfer_loss_inter_class"):
        mh.transfer_loss_inter_class = _CosEmb()
    # attention dropout draws RNG; parity fixtures pin it off (p=0)
    mh.multihead_attn.dropout.p = dropout
    mh.multihead_attn.dot_product_attention.dropout.p = dropout
    return model


def make_targets(boxes_per_img, labels_per_img, image_hw):
    setup()
    from fcos_core.structures.bounding_box import BoxList
    out = []
    for b, l in zip(boxes_per_img, labels_per_img):
        bl = BoxList(torch.as_tensor(b, dtype=torch.float32), (image_hw[1], image_hw[0]), mode="xyxy")
        bl.add_field("labels", torch.as_tensor(l, dtype=torch.int64))
        out.append(bl)
    return out


def forward_detector(cfg, model, images, targets=None, mode="source", forward_target=False):
    """Restates engine/trainer.py:20-72 (importing it would pull fcos_core.data)."""
    from fcos_core.structures.image_list import to_image_list
    images = to_image_list(images)
    features = model["backbone"](images.tensors)
    losses = {}
    features, loss_graph, loss_act, act_maps = model["middle_head"](
        images, features, targets=targets, return_maps=True, mode=mode, forward_target=forward_target)
    if loss_graph is not None:
        node_loss, consistency_loss = loss_graph
        if consistency_loss:
            losses["consistency_loss"] = consistency_loss
        if node_loss:
            losses["node_loss"] = node_loss
    if loss_act is not None:
        losses["act_loss"] = loss_act
    proposals, proposal_losses, _ = model["fcos"](images, features, targets=targets, return_maps=True,
                                                  act_maps=act_maps)
    names = ["P3", "P4", "P5", "P6", "P7"]
    f = {n: features[i] for i, n in enumerate(names)}
    a = {n: act_maps[i] for i, n in enumerate(names)}
    if model["fcos"].training:
        losses.update(proposal_losses)
        return losses, f, a
    return proposals


def da_iteration(cfg, model, images_s, targets_s, images_t, forward_target=False):
    """Three-phase DA iteration, engine/trainer.py:266-385, without optimizers.
    Returns the merged loss dict (python floats); gradients are left in .grad."""
    lam = cfg.MODEL.ADV.CON_DIS_LAMBDA
    out = {}
    for m in model.values():
        m.train()
        m.zero_grad()
    loss_dict, feat_s, maps_s = forward_detector(cfg, model, images_s, targets_s, mode="source")
    loss_dict = {k + "_gs": v for k, v in loss_dict.items()}
    sum(loss_dict.values()).backward(retain_graph=True)
    out.update({k: float(v) for k, v in loss_dict.items()})
    ld = {"zeros": 0 * loss_dict["node_loss_gs"]}
    for lvl in ("P7", "P6", "P5", "P4", "P3"):
        ld["loss_adv_%s_CON_ds" % lvl] = lam * model["dis_%s_CON" % lvl](feat_s[lvl], 1.0, maps_s[lvl], domain="source")
    sum(ld.values()).backward()
    out.update({k: float(v) for k, v in ld.items() if k != "zeros"})
    loss_dict, feat_t, maps_t = forward_detector(cfg, model, images_t, None, mode="target",
                                                 forward_target=forward_target)
    ld = {k + "_gt": v for k, v in loss_dict.items()}
    for lvl in ("P7", "P6", "P5", "P4", "P3"):
        ld["loss_adv_%s_CON_dt" % lvl] = lam * model["dis_%s_CON" % lvl](feat_t[lvl], 0.0, maps_t[lvl], domain="target")
    sum(ld.values()).backward()
    out.update({k: float(v) for k, v in ld.items()})
    return out
