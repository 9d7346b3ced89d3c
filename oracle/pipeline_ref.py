"""oracle/pipeline_ref.py -- numpy restatement of the reference's input pipeline.

TEST INFRASTRUCTURE, NOT PRODUCT CODE (only tests/ import this file).

Restates, per function, what the reference does between a decoded image and the batch the detector receives:
  * Resize.get_size                       fcos_core/data/transforms/transforms.py:34-55
  * F.resize on a PIL image               third-party, ABSENT from /root/reference: torchvision (unpinned by the
                                          reference's requirements; PIL path = Image.resize(size[::-1], BILINEAR) in
                                          every release) -> Pillow's ImagingResample (src/libImaging/Resample.c:
                                          precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc,
                                          ImagingResampleVertical_8bpc); the Pillow installed here is 12.2.0
  * F.hflip / BoxList.transpose           transforms.py:64-73, structures/bounding_box.py:129-165
  * F.to_tensor, Normalize                transforms.py:76-90
  * BoxList.resize                        structures/bounding_box.py:91-127
  * BatchCollator / to_image_list         data/collate_batch.py:5-20, structures/image_list.py:29-72

Pinning: tests/golden/pipeline.npz was produced by the reference's own Compose / BatchCollator classes running on the
real PIL (oracle/make_golden.py gen_pipeline; torchvision's four functional ops restated in ref_harness.setup), and
tests/test_pipeline.py additionally compares resize() with PIL itself on random sizes wherever PIL is importable.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def get_size(image_size, min_size, max_size):
    """Resize.get_size with a single min_size (random.choice of a 1-tuple)."""
    w, h = image_size
    size = min_size
    if max_size is not None:
        mn, mx = float(min((w, h))), float(max((w, h)))
        if mx / mn * size > max_size:
            size = int(round(max_size * mn / mx))
    if (w <= h and w == size) or (h <= w and h == size):
        return (h, w)
    if w < h:
        return (int(size * h / w), size)
    return (size, int(size * w / h))


def coeffs(in_size, out_size):
    """precompute_coeffs (bilinear_filter, support 1.0, box = whole axis) + normalize_coeffs_8bpc."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds, kk = [], []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = []
        for x in range(xmax):
            v = abs((x + xmin - center + 0.5) * ss)
            w.append(1.0 - v if v < 1.0 else 0.0)
        ww = 0.0
        for v in w:
            ww += v
        w = [v / ww if ww != 0.0 else v for v in w]
        kk.append([int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS)) for v in w]
                  + [0] * (ksize - xmax))
        bounds.append((xmin, xmax))
    return bounds, kk


def _resample(img, out_size, axis):
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    bounds, kk = coeffs(src.shape[0], out_size)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, cnt = bounds[xx]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(cnt):
            acc += src[xmin + x] * kk[xx][x]
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis)


def resize(img, oh, ow):
    """Image.resize((ow, oh), BILINEAR) of uint8 [H, W, 3]: horizontal pass first (uint8 intermediate), then vertical."""
    out = img
    if ow != img.shape[1]:
        out = _resample(out, ow, 1)
    if oh != img.shape[0]:
        out = _resample(out, oh, 0)
    return out.copy()


def to_tensor_normalize(img, mean, std, to_bgr255=True):
    """F.to_tensor -> [[2,1,0]] * 255 -> F.normalize, every step rounded to fp32 like torch."""
    t = np.ascontiguousarray(img.transpose(2, 0, 1)).astype(np.float32) / np.float32(255)
    if to_bgr255:
        t = t[[2, 1, 0]] * np.float32(255)
    m = np.asarray(mean, np.float32)[:, None, None]
    s = np.asarray(std, np.float32)[:, None, None]
    return ((t - m) / s).astype(np.float32)


def resize_boxes(boxes, old_size, new_size):
    rw, rh = (float(s) / float(so) for s, so in zip(new_size, old_size))
    if rw == rh:
        return boxes * np.float32(rw)
    return boxes * np.asarray([rw, rh, rw, rh], np.float32)


def hflip_boxes(boxes, width):
    out = boxes.copy()
    out[:, 0] = width - boxes[:, 2] - 1
    out[:, 2] = width - boxes[:, 0] - 1
    return out


def collate(tensors, size_divisible):
    h = max(t.shape[1] for t in tensors)
    w = max(t.shape[2] for t in tensors)
    if size_divisible > 0:
        h = int(math.ceil(h / size_divisible) * size_divisible)
        w = int(math.ceil(w / size_divisible) * size_divisible)
    out = np.zeros((len(tensors), 3, h, w), np.float32)
    for o, t in zip(out, tensors):
        o[:, :t.shape[1], :t.shape[2]] = t
    return out, [t.shape[-2:] for t in tensors]
