"""Input pipeline on the device: the reference's transforms and batch collation (fcos_core/data/transforms/
transforms.py:9-90, data/transforms/build.py:5-44, data/collate_batch.py:5-20) with the same class names, constructor
arguments and call convention ``image, target = transform(image, target)``.

What differs is where the bytes live.  The reference decodes to a PIL image on the host and runs torchvision's PIL ops
there; here a decoded frame is a uint8 [H, W, 3] RGB tensor on the GPU (``U8Image``) and every per-pixel step is a HIP
kernel (csrc/imgproc.hip): Pillow's fixed-point bilinear resampler bit for bit, and one fused
ToTensor + BGR255 + Normalize (+ flip) kernel that writes straight into the collator's zero-padded batch -- as the NCHW
tensor of the reference's ImageList and as the NHWC4 rows the first convolution reads, so the collation is not a pass of
its own.  Targets are (boxes [G,4] xyxy, labels [G]) pairs; BoxList.resize / transpose semantics
(structures/bounding_box.py:91-165) are applied to the boxes.

JPEG decode stays on the host (PIL / any decoder): ``U8Image.from_pil`` / ``from_numpy`` upload the decoded bytes.
"""
import ctypes
import math
import random

import numpy as np
import torch

from . import _lib, ops
from .structures import ImageList

PRECISION_BITS = 32 - 8 - 2  # Pillow Resample.c


def bilinear_tables(in_size, out_size):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc (src/libImaging/Resample.c) for the BILINEAR filter (support
    1.0) over the whole axis: bounds int32 [out][2] = (first input index, taps), coef int32 [out][ksize] with 22
    fractional bits.  Pure double-precision arithmetic in the C code's operation order, so the integers are the
    ones Pillow computes."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    coef = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    one = float(1 << PRECISION_BITS)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [0.0] * xmax
        ww = 0.0
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            if v < 0.0:
                v = -v
            w[x] = 1.0 - v if v < 1.0 else 0.0
            ww += w[x]
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            coef[xx, x] = int(-0.5 + k * one) if k < 0 else int(0.5 + k * one)
        bounds[xx] = (xmin, xmax)
    return bounds, coef, ksize


class U8Image:
    """A decoded RGB frame on the device: uint8 [H, W, 3].  ``size`` is (w, h) like PIL's."""

    def __init__(self, data, flipped=False):
        if data.dtype != torch.uint8 or data.dim() != 3 or data.shape[2] != 3:
            raise ValueError("U8Image needs a uint8 [H, W, 3] tensor")
        if not data.is_cuda:
            raise RuntimeError("scan_amd.data runs only on the GPU (HIP); no CPU fallback")
        self.data = data.contiguous()
        self.flipped = flipped  # RandomHorizontalFlip is applied by the kernel that reads the bytes next

    @property
    def size(self):
        return (self.data.shape[1], self.data.shape[0])

    @staticmethod
    def from_numpy(arr, device="cuda"):
        return U8Image(torch.from_numpy(np.ascontiguousarray(arr)).to(device))

    @staticmethod
    def from_pil(img, device="cuda"):
        return U8Image.from_numpy(np.asarray(img.convert("RGB")), device)

    def materialize(self):
        """uint8 [H, W, 3] with a pending flip applied (what F.hflip returns)."""
        return torch.flip(self.data, dims=(1,)) if self.flipped else self.data


_table_cache = {}


def _tables(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    t = _table_cache.get(key)
    if t is None:
        b, c, k = bilinear_tables(in_size, out_size)
        t = (torch.from_numpy(b).to(device), torch.from_numpy(c).to(device), k)
        if len(_table_cache) > 256:
            _table_cache.clear()
        _table_cache[key] = t
    return t


def resize_u8(image, oh, ow):
    """PIL Image.resize((ow, oh), BILINEAR) of a uint8 [H, W, 3] device tensor."""
    h, w = image.shape[:2]
    dev = image.device
    dst = torch.empty((oh, ow, 3), dtype=torch.uint8, device=dev)
    xb = xc = yb = yc = None
    kx = ky = 0
    if ow != w:
        xb, xc, kx = _tables(w, ow, dev)
    if oh != h:
        yb, yc, ky = _tables(h, oh, dev)
    tmp = torch.empty((h, ow, 3), dtype=torch.uint8, device=dev) if (ow != w and oh != h) else None
    _lib.call("scan_resize_bilinear_u8", ops._ptr(image), h, w, ops._ptr(tmp), ops._ptr(dst), oh, ow, ops._ptr(xb),
              ops._ptr(xc), kx, ops._ptr(yb), ops._ptr(yc), ky, ops._stream())
    return dst


def _f3(v):
    return (ctypes.c_float * 3)(*[float(x) for x in v])


def normalize_into(image, dst, hp, wp, mean, std, to_bgr255, layout):
    """fused ToTensor + Normalize (+ pending flip) of a U8Image into a zero-padded hp x wp slot (layout 0: CHW planes,
    1: NHWC4 rows)."""
    h, w = image.data.shape[:2]
    _lib.call("scan_normalize_image_u8", ops._ptr(image.data), h, w, int(image.flipped), int(bool(to_bgr255)), _f3(mean),
              _f3(std), ops._ptr(dst), hp, wp, layout, ops._stream())


# ----------------------------------------------------------------------------- transforms (reference names)
class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, image, target):
        for t in self.transforms:
            image, target = t(image, target)
        return image, target


def scaled_size(h, w, short_target, long_cap=None):
    """(oh, ow) of a frame whose SHORTER side is brought to ``short_target`` with the aspect ratio kept, unless that would
    push the longer side past ``long_cap`` -- then the target shrinks so the longer side lands on the cap.  The other side
    is truncated, not rounded, and a frame whose shorter side already has the target size is left alone: the integer results
    of the reference's Resize (data/transforms/transforms.py:27-55), pinned by test_get_size_matches_reference_table."""
    lo, hi = (w, h) if w <= h else (h, w)
    t = short_target
    if long_cap is not None and float(hi) / float(lo) * t > long_cap:
        t = int(round(long_cap * float(lo) / float(hi)))
    if lo == t:
        return h, w
    other = int(t * hi / lo)
    return (other, t) if w < h else (t, other)


class Resize:
    """Resize(min_size, max_size) of the reference's transform list: one of ``min_size`` is drawn per frame (multi-scale
    training), boxes follow the frame."""

    def __init__(self, min_size, max_size):
        self.min_size = tuple(min_size) if isinstance(min_size, (list, tuple)) else (min_size,)
        self.max_size = max_size

    def get_size(self, image_size):
        w, h = image_size
        return scaled_size(h, w, random.choice(self.min_size), self.max_size)

    def __call__(self, image, target):
        oh, ow = self.get_size(image.size)
        w, h = image.size
        out = U8Image(resize_u8(image.data, oh, ow), image.flipped)
        if target is not None:
            target = resize_boxes(target, (w, h), out.size)
        return out, target


def resize_boxes(target, old_size, new_size):
    """BoxList.resize (structures/bounding_box.py:91-127): sizes are (w, h)."""
    boxes, labels = target
    rw, rh = (float(s) / float(so) for s, so in zip(new_size, old_size))
    if rw == rh:
        return boxes * rw, labels
    scale = boxes.new_tensor([rw, rh, rw, rh])
    return boxes * scale, labels


def hflip_boxes(target, image_width):
    """BoxList.transpose(FLIP_LEFT_RIGHT) (bounding_box.py:129-165): TO_REMOVE = 1."""
    boxes, labels = target
    xmin, ymin, xmax, ymax = boxes.unbind(-1)
    return torch.stack([image_width - xmax - 1, ymin, image_width - xmin - 1, ymax], -1), labels


class RandomHorizontalFlip:
    def __init__(self, prob=0.5):
        self.prob = prob

    def __call__(self, image, target):
        if random.random() < self.prob:
            image = U8Image(image.data, not image.flipped)
            if target is not None:
                target = hflip_boxes(target, image.size[0])
        return image, target


class ToTensor:
    """F.to_tensor is fused into Normalize's kernel (uint8 -> float / 255); on its own it yields the CHW float tensor."""

    def __call__(self, image, target):
        return image, target


class Normalize:
    def __init__(self, mean, std, to_bgr255=True):
        self.mean, self.std, self.to_bgr255 = list(mean), list(std), to_bgr255

    def __call__(self, image, target):
        return NormalizedImage(image, self), target


class NormalizedImage:
    """The result of the per-image transforms: the bytes plus what is still to be applied by ONE kernel launch -- into
    a tensor of its own (``tensor()``: the reference's [3, h, w]) or into its slot of the collated batch."""

    def __init__(self, image, norm):
        self.image, self.norm = image, norm

    @property
    def shape(self):
        h, w = self.image.data.shape[:2]
        return (3, h, w)

    def tensor(self):
        _, h, w = self.shape
        out = torch.empty((3, h, w), dtype=torch.float32, device=self.image.data.device)
        normalize_into(self.image, out, h, w, self.norm.mean, self.norm.std, self.norm.to_bgr255, 0)
        return out


def build_transforms(cfg, is_train=True):
    """reference data/transforms/build.py:5-44; cfg = scan_amd.config.Cfg."""
    if is_train:
        if cfg.INPUT.MIN_SIZE_RANGE_TRAIN[0] == -1:
            min_size = cfg.INPUT.MIN_SIZE_TRAIN
        else:
            assert len(cfg.INPUT.MIN_SIZE_RANGE_TRAIN) == 2, \
                "MIN_SIZE_RANGE_TRAIN must have two elements (lower bound, upper bound)"
            min_size = list(range(cfg.INPUT.MIN_SIZE_RANGE_TRAIN[0], cfg.INPUT.MIN_SIZE_RANGE_TRAIN[1] + 1))
        max_size = cfg.INPUT.MAX_SIZE_TRAIN
        flip_prob = 0.5
    else:
        min_size = cfg.INPUT.MIN_SIZE_TEST
        max_size = cfg.INPUT.MAX_SIZE_TEST
        flip_prob = 0
    return Compose([Resize(min_size, max_size), RandomHorizontalFlip(flip_prob), ToTensor(),
                    Normalize(cfg.INPUT.PIXEL_MEAN, cfg.INPUT.PIXEL_STD, cfg.INPUT.TO_BGR255)])


class BatchCollator:
    """reference data/collate_batch.py:5-20: (images, targets, ids) of a list of samples; the images become an ImageList
    zero-padded to a multiple of size_divisible.  Each NormalizedImage is written by its one kernel launch directly
    into the batch: ``tensors`` [N, 3, Hp, Wp] (the reference layout) and, with rows=True, also ``rows`` [N*Hp*Wp, 4] +
    ``shape``, which engine.forward_detector hands to the backbone without the NCHW -> NHWC pass."""

    def __init__(self, size_divisible=0, rows=True, nchw=True):
        self.size_divisible = size_divisible
        self.rows, self.nchw = rows, nchw

    def __call__(self, batch):
        images, targets, ids = list(zip(*batch))
        h = max(im.shape[1] for im in images)
        w = max(im.shape[2] for im in images)
        if self.size_divisible > 0:
            h = int(math.ceil(h / self.size_divisible) * self.size_divisible)
            w = int(math.ceil(w / self.size_divisible) * self.size_divisible)
        dev = images[0].image.data.device
        n = len(images)
        tensors = torch.empty((n, 3, h, w), dtype=torch.float32, device=dev) if self.nchw else None
        rows = torch.empty((n * h * w, 4), dtype=torch.float32, device=dev) if self.rows else None
        for i, im in enumerate(images):
            nm = im.norm
            if tensors is not None:
                normalize_into(im.image, tensors[i], h, w, nm.mean, nm.std, nm.to_bgr255, 0)
            if rows is not None:
                normalize_into(im.image, rows[i * h * w:(i + 1) * h * w], h, w, nm.mean, nm.std, nm.to_bgr255, 1)
        il = ImageList(tensors, [im.shape[-2:] for im in images])
        if rows is not None:
            il.rows, il.shape = rows, ops.PyramidShape(n, [(h, w)])
        return il, targets, ids
