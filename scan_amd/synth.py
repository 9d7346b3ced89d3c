"""Procedural (seeded, network-free) weights and synthetic inputs.

There is no network for ImageNet weights or Cityscapes frames, so the bench and
every parity test use these generators.  They are keyed by the reference's
``state_dict`` names so the same tensors can be loaded into the
reference modules (oracle/make_golden.py), the CPU restatement
(oracle/scan_ref.py) and the HIP modules (scan_amd/modeling).

Input synthesis follows SURVEY.md 8d: frames are U[0,255) minus
INPUT.PIXEL_MEAN in BGR order, std 1 (reference fcos_core/config/defaults.py:57-61,
fcos_core/data/transforms/transforms.py:80-90); source ground truth is 12 boxes
per image with sqrt(area) log-uniform in [16, 512] px so that all five FCOS
size-of-interest ranges (fcos_core/modeling/rpn/fcos/loss.py:41-47) are hit.
"""
import math
import zlib

import numpy as np
import torch

PIXEL_MEAN = (102.9801, 115.9465, 122.7717)

VGG_CONV_IDX = (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)
VGG_CHANNELS = ((3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 256),
                (256, 512), (512, 512), (512, 512), (512, 512), (512, 512), (512, 512))


def _rs(key):
    return np.random.RandomState(zlib.crc32(key.encode()) & 0x7FFFFFFF)


def _normal(key, shape, std):
    return torch.from_numpy((_rs(key).standard_normal(shape) * std).astype(np.float32))


def _uniform(key, shape, bound):
    return torch.from_numpy(_rs(key).uniform(-bound, bound, shape).astype(np.float32))


def _conv(sd, name, cout, cin, k, gain, bias_std=0.02, kw=None):
    kh, kwid = (k, k) if kw is None else (k, kw)
    fan_in = cin * kh * kwid
    sd[name + ".weight"] = _normal(name + ".weight", (cout, cin, kh, kwid), gain / math.sqrt(fan_in))
    sd[name + ".bias"] = _normal(name + ".bias", (cout,), bias_std)


def _gn(sd, name, c):
    sd[name + ".weight"] = 1.0 + _normal(name + ".weight", (c,), 0.1)
    sd[name + ".bias"] = _normal(name + ".bias", (c,), 0.1)


def _linear(sd, name, cout, cin, gain=1.0):
    sd[name + ".weight"] = _normal(name + ".weight", (cout, cin), gain / math.sqrt(cin))
    sd[name + ".bias"] = _normal(name + ".bias", (cout,), 0.02)


def _tower(sd, prefix, n, c=256):
    for i in range(n):
        _conv(sd, "%s.%d" % (prefix, 3 * i), c, c, 3, math.sqrt(2.0))
        _gn(sd, "%s.%d" % (prefix, 3 * i + 1), c)


def backbone_state_dict():
    """VGG16 body + FPN(P3..P5) + P6/P7  (reference backbone/backbone.py:21-44)."""
    sd = {}
    for idx, (cin, cout) in zip(VGG_CONV_IDX, VGG_CHANNELS):
        _conv(sd, "body.features.%d" % idx, cout, cin, 3, math.sqrt(2.0))
    for lvl, cin in ((3, 256), (4, 512), (5, 512)):
        _conv(sd, "fpn.fpn_inner%d" % lvl, 256, cin, 1, 1.0)
        _conv(sd, "fpn.fpn_layer%d" % lvl, 256, 256, 3, 1.0)
    _conv(sd, "fpn.top_blocks.p6", 256, 256, 3, 1.0)
    _conv(sd, "fpn.top_blocks.p7", 256, 256, 3, math.sqrt(2.0))
    return sd


RESNET_BLOCKS = {"R-50": (3, 4, 6, 3), "R-101": (3, 4, 23, 3)}


def _conv_nobias(sd, name, cout, cin, k, gain):
    sd[name + ".weight"] = _normal(name + ".weight", (cout, cin, k, k), gain / math.sqrt(cin * k * k))


def _frozen_bn(sd, name, c, gain=1.0):
    """FrozenBatchNorm2d buffers (reference layers/batch_norm.py:5-24); running_var strictly positive (no eps there)."""
    sd[name + ".weight"] = gain * (1.0 + _normal(name + ".weight", (c,), 0.1))
    sd[name + ".bias"] = _normal(name + ".bias", (c,), 0.1)
    sd[name + ".running_mean"] = _normal(name + ".running_mean", (c,), 0.1)
    sd[name + ".running_var"] = torch.from_numpy(_rs(name + ".running_var").uniform(0.5, 1.5, (c,)).astype(np.float32))


def resnet_backbone_state_dict(arch="R-50"):
    """ResNet body + FPN(P3..P5) + P6/P7 of R-50/101-FPN-RETINANET (reference backbone/backbone.py:94-117,
    backbone/resnet.py).  The last BN of every block is scaled down so 16 residual joins keep O(1) activations."""
    sd = {}
    _conv_nobias(sd, "body.stem.conv1", 64, 3, 7, math.sqrt(2.0) / 60.0)  # inputs are O(70): bring them to O(1)
    _frozen_bn(sd, "body.stem.bn1", 64)
    cin = 64
    for i, n in enumerate(RESNET_BLOCKS[arch], 1):
        mid, cout = 64 * 2 ** (i - 1), 256 * 2 ** (i - 1)
        for b in range(n):
            p = "body.layer%d.%d" % (i, b)
            if cin != cout:
                _conv_nobias(sd, p + ".downsample.0", cout, cin, 1, 1.0)
                _frozen_bn(sd, p + ".downsample.1", cout)
            _conv_nobias(sd, p + ".conv1", mid, cin, 1, math.sqrt(2.0))
            _frozen_bn(sd, p + ".bn1", mid)
            _conv_nobias(sd, p + ".conv2", mid, mid, 3, math.sqrt(2.0))
            _frozen_bn(sd, p + ".bn2", mid)
            _conv_nobias(sd, p + ".conv3", cout, mid, 1, math.sqrt(2.0))
            _frozen_bn(sd, p + ".bn3", cout, gain=0.5)
            cin = cout
    for idx, c in ((2, 512), (3, 1024), (4, 2048)):
        _conv(sd, "fpn.fpn_inner%d" % idx, 256, c, 1, 1.0)
        _conv(sd, "fpn.fpn_layer%d" % idx, 256, 256, 3, 1.0)
    _conv(sd, "fpn.top_blocks.p6", 256, 256, 3, 1.0)
    _conv(sd, "fpn.top_blocks.p7", 256, 256, 3, math.sqrt(2.0))
    return sd


def middle_head_state_dict(num_classes=9, proto_iter=3):
    """GRAPHModule (reference rpn/fcos/condgraph.py:127-253)."""
    K = num_classes
    sd = {"prototype": _normal("prototype", (K, 256, proto_iter), 1.0)}
    _tower(sd, "head_in.middle_tower", 2)
    _conv(sd, "head_out.middle_tower.0", 256, 256 + K, 3, math.sqrt(2.0))
    _linear(sd, "proto_cls_hidden", 512, 256)
    _linear(sd, "proto_cls", K, 512)
    for n in ("linear_k", "linear_v", "linear_q", "linear_final"):
        _linear(sd, "multihead_attn." + n, 256, 256)
    sd["multihead_attn.layer_norm.weight"] = 1.0 + _normal("mha.ln.w", (256,), 0.1)
    sd["multihead_attn.layer_norm.bias"] = _normal("mha.ln.b", (256,), 0.1)
    _conv(sd, "cond_nx1", 256, 512, proto_iter, 1.0, kw=1)
    bound = 1.0 / math.sqrt(512)
    for layer, cin in ((0, 256), (1, 512)):
        sd["cond_rnn.weight_ih_l%d" % layer] = _uniform("rnn.wih%d" % layer, (512, cin), bound)
        sd["cond_rnn.weight_hh_l%d" % layer] = _uniform("rnn.whh%d" % layer, (512, 512), bound)
        sd["cond_rnn.bias_ih_l%d" % layer] = _uniform("rnn.bih%d" % layer, (512,), bound)
        sd["cond_rnn.bias_hh_l%d" % layer] = _uniform("rnn.bhh%d" % layer, (512,), bound)
    _linear(sd, "cond_2", 256, 512)
    return sd


def fcos_state_dict(num_classes=9):
    """FCOSHead (reference rpn/fcos/fcos.py:13-87)."""
    sd = {}
    _tower(sd, "head.cls_tower", 4)
    _tower(sd, "head.bbox_tower", 4)
    _conv(sd, "head.cls_logits", num_classes - 1, 256, 3, 0.5)
    sd["head.cls_logits.bias"] = sd["head.cls_logits.bias"] - math.log(99.0)
    _conv(sd, "head.bbox_pred", 4, 256, 3, 0.5)
    sd["head.bbox_pred.bias"] = sd["head.bbox_pred.bias"] + 2.0
    _conv(sd, "head.centerness", 1, 256, 3, 0.5)
    for i in range(5):
        sd["head.scales.%d.scale" % i] = torch.tensor([1.0 + 0.05 * i], dtype=torch.float32)
    return sd


def discriminator_state_dict(level, num_classes=9):
    """FCOSDiscriminator_con (reference discriminator/fcos_head_discriminator_con.py:12-87)."""
    sd = {}
    tag = "dis_%s" % level
    t = {}
    _tower(t, tag + ".dis_tower", 4)
    for c in range(num_classes - 1):
        _conv(t, "%s.classifier_cls_%d.0" % (tag, c), 128, 257, 3, math.sqrt(2.0))
        _conv(t, "%s.classifier_cls_%d.2" % (tag, c), 1, 128, 3, 1.0)
    for k, v in t.items():
        sd[k[len(tag) + 1:]] = v
    return sd


LEVEL_NAMES = ("P3", "P4", "P5", "P6", "P7")


def all_state_dicts(num_classes=9, conv_body="VGG-16-FPN-RETINANET"):
    out = {
        "backbone": backbone_state_dict() if conv_body.startswith("VGG") else
        resnet_backbone_state_dict(conv_body[:-len("-FPN-RETINANET")]),
        "middle_head": middle_head_state_dict(num_classes),
        "fcos": fcos_state_dict(num_classes),
    }
    for lvl in LEVEL_NAMES:
        out["dis_%s_CON" % lvl] = discriminator_state_dict(lvl, num_classes)
    return out


def synth_images(n, h, w, seed0=1234):
    """[n,3,h,w] fp32, U[0,255) - PIXEL_MEAN (BGR), one generator seed per image."""
    imgs = []
    mean = np.asarray(PIXEL_MEAN, dtype=np.float32).reshape(3, 1, 1)
    for i in range(n):
        rs = np.random.RandomState(seed0 + i)
        imgs.append(torch.from_numpy(rs.uniform(0.0, 255.0, (3, h, w)).astype(np.float32) - mean))
    return torch.stack(imgs, 0)


def synth_image_list(sizes, seed0=1234):
    """ragged batch: list of [3,h_i,w_i] images (same per-image generator as synth_images)."""
    return [synth_images(1, h, w, seed0 + i)[0] for i, (h, w) in enumerate(sizes)]


def synth_targets(n, h, w, num_fg=8, boxes_per_img=12, seed0=4321):
    """Per image: (boxes [G,4] xyxy fp32, labels [G] int64 in 1..num_fg)."""
    out = []
    smax = min(512.0, 0.9 * min(h, w))
    smin = min(16.0, smax / 4)
    for i in range(n):
        rs = np.random.RandomState(seed0 + i)
        side = np.exp(rs.uniform(math.log(smin), math.log(smax), boxes_per_img))
        aspect = rs.uniform(0.5, 2.0, boxes_per_img)
        bw = np.minimum(side * np.sqrt(aspect), w - 2.0)
        bh = np.minimum(side / np.sqrt(aspect), h - 2.0)
        cx = rs.uniform(bw / 2, w - 1 - bw / 2)
        cy = rs.uniform(bh / 2, h - 1 - bh / 2)
        boxes = np.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1).astype(np.float32)
        labels = rs.randint(1, num_fg + 1, boxes_per_img).astype(np.int64)
        out.append((torch.from_numpy(boxes), torch.from_numpy(labels)))
    return out


# ----------------------------------------------------------------------------- fixture inputs shared by the golden
# generator (oracle/make_golden.py, run against the reference) and the parity tests
TRAJ_ABSENT = (1, 5)  # trajectory iteration 1 holds no box of class 5 (rewritten to 6): the 'exists' mask path


def traj_batch(it, H, W, N, K):
    """inputs of trajectory iteration `it` (a different batch every iteration)."""
    imgs_s = synth_images(N, H, W, 1234 + 17 * it)
    imgs_t = synth_images(N, H, W, 2234 + 17 * it)
    tg = synth_targets(N, H, W, K - 1, 12, 4321 + 17 * it)
    if it == TRAJ_ABSENT[0]:
        tg = [(b, torch.where(l == TRAJ_ABSENT[1], l + 1, l)) for b, l in tg]
    return imgs_s, tg, imgs_t


# procedural head biases moved so that every test mode yields detections: sigmoid(cls) passes INFERENCE_TH and
# neighbouring locations predict boxes that overlap above NMS_TH (fixtures inference2_*)
INF2_SHIFT = {"cls_bias": 3.0, "bbox_bias": 1.0}


def shifted_state_dicts(K, conv_body="VGG-16-FPN-RETINANET"):
    sds = all_state_dicts(K, conv_body)
    sds["fcos"]["head.cls_logits.bias"] = sds["fcos"]["head.cls_logits.bias"] + INF2_SHIFT["cls_bias"]
    sds["fcos"]["head.bbox_pred.bias"] = sds["fcos"]["head.bbox_pred.bias"] + INF2_SHIFT["bbox_bias"]
    return sds


def const_of(name):
    """per-tensor constant of the reference-written checkpoint fixture (tests/golden/refckpt_*.pth.gz: constant
    tensors gzip to kilobytes and still tell tensors apart)."""
    return ((zlib.crc32(name.encode()) % 2000) - 1000) / 4096.0


def synth_u8_image(h, w, seed):
    """uint8 RGB test frame [h, w, 3] with structure at several scales (smooth gradients + blocks + noise), so that a
    resize with wrong coefficients, bounds or rounding shows up in many pixels."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(xx * 255.0 / max(w - 1, 1)), (yy * 255.0 / max(h - 1, 1)), ((xx + yy) % 64) * 4.0], -1)
    blocks = rs.randint(0, 256, (h // 8 + 1, w // 8 + 1, 3)).repeat(8, 0).repeat(8, 1)[:h, :w]
    noise = rs.randint(0, 256, (h, w, 3))
    return ((base + blocks + noise) / 3.0).astype(np.uint8)
