"""ResNet-50/101 + FPN (P3..P7) backbone on pyramid activations ("R-50-FPN-RETINANET", BASELINE.json configs[3]).

Mirrors the reference's build_resnet_fpn_p3p7_backbone (fcos_core/modeling/backbone/backbone.py:94-117):
ResNet body (backbone/resnet.py:80-145: StemWithFixedBatchNorm, BottleneckWithFixedBatchNorm, STRIDE_IN_1X1 True,
stages (3, 4, 6, 3) / (3, 4, 23, 3), FREEZE_CONV_BODY_AT 2 = stem + layer1 frozen), FPN on C3..C5 with
RESNETS.BACKBONE_OUT_CHANNELS 256 (backbone/fpn.py:44-91) and LastLevelP6P7 fed by P5 (RETINANET.USE_C5 False), i.e.
the settings of the reference's own ResNet yamls (configs/epm/*R_101*).  ``state_dict`` keys equal the reference's
(body.stem.conv1.weight, body.stem.bn1.{weight,bias,running_mean,running_var}, body.layerL.B.convK.weight,
body.layerL.B.bnK.*, body.layerL.0.downsample.{0.weight,1.*}, fpn.fpn_inner{2,3,4}.*, fpn.fpn_layer{2,3,4}.*,
fpn.top_blocks.{p6,p7}.*).

FrozenBatchNorm2d (layers/batch_norm.py:5-24: y = x * w * rsqrt(var) + (b - mean * w * rsqrt(var)), no eps) is an
affine per output channel of the conv in front of it: it is folded into that conv's weights and bias, so conv + BN
(+ ReLU) is ONE launch of the conv kernels (3x3/stride 1 on the bf16x3 matrix-core kernel, 1x1 / stride 2 / 7x7 on the
fp32 implicit-GEMM kernel).  DCNv2: the reference has no source for it (layers/misc.py:135-141 imports an absent
package; USE_DCN_IN_TOWER False) -- not built.
"""
import torch
from torch import nn

from .. import ops
from ..ops import PyramidShape
from .backbone import LastLevelP6P7, conv_holder, fpn_top_down

STAGE_BLOCKS = {"R-50": (3, 4, 6, 3), "R-101": (3, 4, 23, 3)}


class FrozenBatchNorm2d(nn.Module):
    """buffers only (reference layers/batch_norm.py:5-24); ``fold`` gives the affine it applies."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))

    def fold(self):
        scale = self.weight * self.running_var.rsqrt()
        return scale, self.bias - self.running_mean * scale


def _conv_nobias(cin, cout, k, stride=1):
    m = nn.Conv2d(cin, cout, k, stride, k // 2, bias=False)
    m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    return m


def conv_bn(rows, shape, conv, bn, relu):
    """conv -> FrozenBN (-> ReLU) as one conv launch with folded weights; returns (rows, out shape)."""
    scale, shift = bn.fold()
    w = conv.weight * scale.view(-1, 1, 1, 1)
    k, s = conv.kernel_size[0], conv.stride[0]
    return ops.conv2d(rows, w, shift, shape, k, s, relu=relu), shape.conv_out(k, s)


class Bottleneck(nn.Module):
    def __init__(self, cin, mid, cout, stride):
        super().__init__()
        self.downsample = None
        if cin != cout:
            self.downsample = nn.Sequential(_conv_nobias(cin, cout, 1, stride), FrozenBatchNorm2d(cout))
        self.conv1 = _conv_nobias(cin, mid, 1, stride)  # STRIDE_IN_1X1 True (resnet.py:264)
        self.bn1 = FrozenBatchNorm2d(mid)
        self.conv2 = _conv_nobias(mid, mid, 3, 1)
        self.bn2 = FrozenBatchNorm2d(mid)
        self.conv3 = _conv_nobias(mid, cout, 1, 1)
        self.bn3 = FrozenBatchNorm2d(cout)

    def forward(self, rows, shape):
        out, s1 = conv_bn(rows, shape, self.conv1, self.bn1, True)
        out, _ = conv_bn(out, s1, self.conv2, self.bn2, True)
        out, _ = conv_bn(out, s1, self.conv3, self.bn3, False)
        identity = rows
        if self.downsample is not None:
            identity, _ = conv_bn(rows, shape, self.downsample[0], self.downsample[1], False)
        return ops.add_relu(out, identity), s1


class Stem(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = _conv_nobias(3, 64, 7, 2)
        self.bn1 = FrozenBatchNorm2d(64)

    def forward(self, rows, shape):
        rows, shape = conv_bn(rows, shape, self.conv1, self.bn1, True)
        return ops.maxpool3x3s2(rows, shape)


class ResNetBody(nn.Module):
    def __init__(self, blocks=(3, 4, 6, 3), freeze_at=2):
        super().__init__()
        self.stem = Stem()
        cin = 64
        for i, n in enumerate(blocks, 1):
            mid, cout = 64 * 2 ** (i - 1), 256 * 2 ** (i - 1)
            layer = []
            for b in range(n):
                layer.append(Bottleneck(cin, mid, cout, (2 if i > 1 else 1) if b == 0 else 1))
                cin = cout
            setattr(self, "layer%d" % i, nn.ModuleList(layer))
        if freeze_at < 1:
            raise ValueError("the stem pooling has no backward: FREEZE_CONV_BODY_AT must be >= 1")
        for idx in range(freeze_at):  # resnet.py:128-138
            m = self.stem if idx == 0 else getattr(self, "layer%d" % idx)
            for p in m.parameters():
                p.requires_grad = False

    def forward(self, rows, shape):
        rows, shape = self.stem(rows, shape)
        outs = []
        for i in range(1, 5):
            for blk in getattr(self, "layer%d" % i):
                rows, shape = blk(rows, shape)
            outs.append((rows, shape))
        return outs


class ResNetFPN(nn.Module):
    """FPN with in_channels_list [0, 512, 1024, 2048] -> blocks named fpn_inner{2,3,4} / fpn_layer{2,3,4}."""

    def __init__(self):
        super().__init__()
        for idx, cin in ((2, 512), (3, 1024), (4, 2048)):
            setattr(self, "fpn_inner%d" % idx, conv_holder(cin, 256, 1))
            setattr(self, "fpn_layer%d" % idx, conv_holder(256, 256, 3))
        self.top_blocks = LastLevelP6P7(256, 256)

    def forward(self, c3, c4, c5):
        return fpn_top_down([getattr(self, "fpn_inner%d" % l) for l in (2, 3, 4)],
                            [getattr(self, "fpn_layer%d" % l) for l in (2, 3, 4)], self.top_blocks, c3, c4, c5)


class ResNetFPNBackbone(nn.Module):
    """model["backbone"]: images [N,3,H,W] (H, W multiples of 32) -> (rows [M,256], PyramidShape of P3..P7)."""
    out_channels = 256

    def __init__(self, arch="R-50", freeze_at=2):
        super().__init__()
        self.body = ResNetBody(STAGE_BLOCKS[arch], freeze_at)
        self.fpn = ResNetFPN()

    def forward(self, images, rows=None, shape=None):
        """images [N,3,H,W]; or rows [N*H*W, 4] + its one-level PyramidShape (data.BatchCollator writes the batch in
        this layout directly: no NCHW -> NHWC pass)."""
        if rows is None:
            if not images.is_cuda:
                raise RuntimeError("scan_amd backbone runs only on the GPU (HIP); no CPU fallback")
            rows, shape = ops.nchw_to_rows(images, 4)
        outs = self.body(rows, shape)
        return self.fpn(outs[1], outs[2], outs[3])


def build_resnet_fpn_backbone(arch="R-50", freeze_at=2):
    return ResNetFPNBackbone(arch, freeze_at)
