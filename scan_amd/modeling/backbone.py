"""VGG16 + FPN (P3..P7) backbone on pyramid activations, every conv on the fp32-MFMA
HIP kernel.  Mirrors the reference's "VGG-16-FPN-RETINANET" builder
(fcos_core/modeling/backbone/backbone.py:21-44): mmcv VGG16 body
(backbone/mmdetection/vgg.py:36-169, stages 1-2 frozen), FPN on C3..C5
(backbone/fpn.py:44-91, plain convs: FPN.USE_GN/USE_RELU False) and
LastLevelP6P7 fed by P5 (fpn.py:105-130, RETINANET.USE_C5 False).
``state_dict`` keys equal the reference's (body.features.N.*, fpn.fpn_inner3.*, ...).
"""
import torch
import torch.nn.functional as F
from torch import nn

from .. import ops
from ..ops import PyramidShape

VGG_STAGES = ((0, 2), (5, 7), (10, 12, 14), (17, 19, 21), (24, 26, 28))
VGG_PLANES = (64, 128, 256, 512, 512)


def conv_holder(cin, cout, k, stride=1):
    """nn.Conv2d used as a parameter container (reference names/shapes), stored channels_last so the
    HIP kernels read it as [Cout][k*k][Cin] without a repack."""
    m = nn.Conv2d(cin, cout, k, stride, k // 2)
    m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    return m


def maxpool2x2(rows, shape, relu_input=False):
    """2x2/2 max-pool of a single-level pyramid (HIP kernel, scan_maxpool2x2_*)."""
    return ops.maxpool2x2(rows, shape, relu_input)


def upsample2x(rows, shape):
    """nearest 2x upsample of a single-level pyramid (F.interpolate(scale_factor=2, 'nearest'))."""
    (h, w), n = shape.sizes[0], shape.n_images
    c = rows.shape[1]
    y = rows.view(n, h, 1, w, 1, c).expand(n, h, 2, w, 2, c).reshape(n * 4 * h * w, c)
    return y


class VGGBody(nn.Module):
    def __init__(self, frozen_stages=2):
        super().__init__()
        layers = []
        cin = 3
        for stage, planes in zip(VGG_STAGES, VGG_PLANES):
            for _ in stage:
                layers += [conv_holder(cin, planes, 3), nn.ReLU(inplace=True)]
                cin = planes
            layers.append(nn.MaxPool2d(2, 2))
        self.features = nn.Sequential(*layers)
        # reference vgg.py:128-138: stages < frozen_stages have requires_grad False
        for s in range(frozen_stages):
            for idx in VGG_STAGES[s]:
                for p in self.features[idx].parameters():
                    p.requires_grad = False

    def forward(self, rows, shape):
        # every conv output here has exactly one consumer (the next conv or the stage's max-pool), so the ReLU
        # backward is folded into that consumer's dgrad / pool-backward epilogue instead of a pass of its own
        outs = []
        self.first_trainable_out = None
        for stage in VGG_STAGES:
            for j, idx in enumerate(stage):
                m = self.features[idx]
                if j == len(stage) - 1 and ops.conv_pool_fusable(rows, m.weight, m.bias, shape):
                    # frozen stage: conv + ReLU + max-pool in one launch, the full-resolution map is never stored
                    rows = ops.conv2d(rows, m.weight, m.bias, shape, 3, 1, relu=True, pool=True)
                    (h, w) = shape.sizes[0]
                    shape = PyramidShape(shape.n_images, [(h // 2, w // 2)])
                    break
                rows = ops.conv2d(rows, m.weight, m.bias, shape, 3, 1, relu="deferred", mask_dx=j > 0)
                if self.first_trainable_out is None and rows.requires_grad:
                    # output of the first conv with a gradient (conv3_1 at frozen_stages = 2): its autograd node is the
                    # LAST backbone node of a backward pass (its input needs no gradient) -- VGG16FPN.grad_marks["node:rest"]
                    self.first_trainable_out = rows
            else:
                rows, shape = maxpool2x2(rows, shape, relu_input=True)
            outs.append((rows, shape))
        return outs


def fpn_top_down(inner, layer, top_blocks, c3, c4, c5):
    """FPN on C3..C5 + LastLevelP6P7 on P5 (reference backbone/fpn.py:44-130); inner / layer: the 1x1 lateral and 3x3
    output conv holders of the three levels, finest first.  -> (rows [M, 256], PyramidShape of P3..P7)."""
    # The five output convs write straight into their row ranges of ONE [M, 256] pyramid matrix (no concatenation
    # afterwards), and a top-down join (lateral + nearest-2x up-sampled coarser map) is one pass.
    s3, s4, s5 = c3[1], c4[1], c5[1]
    s6 = s5.conv_out(3, 2)
    s7 = s6.conv_out(3, 2)
    shape = PyramidShape(s3.n_images, [s.sizes[0] for s in (s3, s4, s5, s6, s7)])
    direct = c3[0].is_cuda and ops.FPN_DIRECT
    buf = c3[0].new_empty((shape.rows, 256)) if direct else None

    def conv(m, rs, k, stride=1, lvl=None):
        out = buf[shape.row_off[lvl]:shape.row_off[lvl + 1]] if (direct and lvl is not None) else None
        return ops.conv2d(rs[0], m.weight, m.bias, rs[1], k, stride, out=out), rs[1].conv_out(k, stride)

    def join(lat, coarse):
        (hl, wl), (hc, wc) = lat[1].sizes[0], coarse[1].sizes[0]
        if direct and (hl, wl) == (2 * hc, 2 * wc):
            return ops.upsample2x_add(lat[0], coarse[0], coarse[1]), lat[1]
        return lat[0] + upsample2x(*coarse), lat[1]

    inner5 = conv(inner[2], c5, 1)
    p5 = conv(layer[2], inner5, 3, lvl=2)
    inner4 = join(conv(inner[1], c4, 1), inner5)
    p4 = conv(layer[1], inner4, 3, lvl=1)
    inner3 = join(conv(inner[0], c3, 1), inner4)
    p3 = conv(layer[0], inner3, 3, lvl=0)
    p6 = conv(top_blocks.p6, p5, 3, 2, lvl=3)
    p7 = conv(top_blocks.p7, (F.relu(p6[0]), p6[1]), 3, 2, lvl=4)
    levels = [p3, p4, p5, p6, p7]
    if direct:
        rows = ops.assemble_rows(buf, [l[0] for l in levels])
    else:
        rows = torch.cat([l[0] for l in levels], 0)
    return rows, shape


class LastLevelP6P7(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.p6 = conv_holder(cin, cout, 3, 2)
        self.p7 = conv_holder(cout, cout, 3, 2)


class FPN(nn.Module):
    def __init__(self):
        super().__init__()
        for lvl, cin in ((3, 256), (4, 512), (5, 512)):
            setattr(self, "fpn_inner%d" % lvl, conv_holder(cin, 256, 1))
            setattr(self, "fpn_layer%d" % lvl, conv_holder(256, 256, 3))
        self.top_blocks = LastLevelP6P7(256, 256)

    def forward(self, c3, c4, c5):
        return fpn_top_down([getattr(self, "fpn_inner%d" % l) for l in (3, 4, 5)],
                            [getattr(self, "fpn_layer%d" % l) for l in (3, 4, 5)], self.top_blocks, c3, c4, c5)


class VGG16FPN(nn.Module):
    """model["backbone"]: images [N,3,H,W] (H, W multiples of 32: structures.to_image_list pads) -> (rows [M,256], PyramidShape of P3..P7)."""
    out_channels = 256

    def __init__(self):
        super().__init__()
        self.body = VGGBody()
        self.fpn = FPN()
        # data parallelism (engine.Trainer): tensors whose gradient marks the point of the backward at
        # which a stage's parameter gradients are final, so their slice of the gradient arena can be all-reduced while
        # the earlier stages still back-propagate.  Filled per forward when record_grad_marks is set (the trainer
        # clears it after the step: the entries keep the autograd graph alive).
        #   "out": the pyramid handed to the middle head -> the middle head's gradients are final
        #   "c4":  stage-4 output (conv5_1 and fpn_inner4 have back-propagated) -> conv5_x weights
        #   "c3":  stage-3 output (conv4_1 and fpn_inner3 have back-propagated) -> conv4_x and every FPN weight
        #   "node:rest": output of the first trainable conv (conv3_1).  No tensor below it receives a gradient (stages 1-2
        #          are frozen), so the trainer hooks its autograd NODE: when that node has run, conv3_1's weight gradient --
        #          the last kernel of the backbone's backward -- is queued and the remaining bucket (conv3_x + every bias)
        #          is final.  Until round 5 that bucket was only issued after backward() had returned.
        self.record_grad_marks = False
        self.grad_marks = {}
        self.grad_stage_params = [
            ("c4", ["body.features.%d.weight" % i for i in VGG_STAGES[4]]),
            ("c3", ["body.features.%d.weight" % i for i in VGG_STAGES[3]]
             + ["fpn.fpn_inner%d.weight" % l for l in (3, 4, 5)] + ["fpn.fpn_layer%d.weight" % l for l in (3, 4, 5)]
             + ["fpn.top_blocks.p6.weight", "fpn.top_blocks.p7.weight"]),
        ]

    def forward(self, images, rows=None, shape=None):
        """images [N,3,H,W]; or rows [N*H*W, 4] + its one-level PyramidShape (data.BatchCollator writes the batch in
        this layout directly: no NCHW -> NHWC pass)."""
        if rows is None:
            if not images.is_cuda:
                raise RuntimeError("scan_amd backbone runs only on the GPU (HIP); no CPU fallback")
            rows, shape = ops.nchw_to_rows(images, 4)
        outs = self.body(rows, shape)
        out = self.fpn(outs[2], outs[3], outs[4])
        if self.record_grad_marks and torch.is_grad_enabled():
            self.grad_marks = {"out": out[0], "c4": outs[3][0], "c3": outs[2][0]}
            if self.body.first_trainable_out is not None:
                self.grad_marks["node:rest"] = self.body.first_trainable_out
        self.body.first_trainable_out = None
        return out


def build_backbone(cfg=None):
    return VGG16FPN()
