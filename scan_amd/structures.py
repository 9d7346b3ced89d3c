"""Batch container semantics the hot path relies on (reference fcos_core/structures/image_list.py:8-72).

The reference collator hands the detector an ImageList: images of different sizes zero-padded at the bottom /
right to a common size that is a multiple of SIZE_DIVISIBILITY (32), plus the true (h, w) of every image, which
inference uses to clip boxes (structures/bounding_box.py:214-224).  The pyramid layout needs nothing else:
padding pixels are ordinary zero-valued pixels of the batch tensor."""
import math

import torch


class ImageList:
    def __init__(self, tensors, image_sizes):
        self.tensors = tensors
        self.image_sizes = [tuple(int(v) for v in s) for s in image_sizes]

    def to(self, *args, **kwargs):
        return ImageList(self.tensors.to(*args, **kwargs), self.image_sizes)


def to_image_list(tensors, size_divisible=0):
    """ImageList | Tensor [N,3,H,W] or [3,H,W] | list of [3,h_i,w_i] -> ImageList."""
    if isinstance(tensors, ImageList):
        return tensors
    if isinstance(tensors, torch.Tensor):
        if size_divisible > 0:
            tensors = [tensors] if tensors.dim() == 3 else list(tensors)
        else:
            if tensors.dim() == 3:
                tensors = tensors[None]
            if tensors.dim() != 4:
                raise ValueError("to_image_list: expected a [N,3,H,W] or [3,H,W] tensor")
            return ImageList(tensors, [t.shape[-2:] for t in tensors])
    if not isinstance(tensors, (list, tuple)):
        raise TypeError("Unsupported type for to_image_list: %s" % type(tensors))
    c = tensors[0].shape[0]
    h = max(t.shape[1] for t in tensors)
    w = max(t.shape[2] for t in tensors)
    if size_divisible > 0:
        h = int(math.ceil(h / size_divisible) * size_divisible)
        w = int(math.ceil(w / size_divisible) * size_divisible)
    batch = tensors[0].new_zeros((len(tensors), c, h, w))
    for t, dst in zip(tensors, batch):
        dst[:, :t.shape[1], :t.shape[2]].copy_(t)
    return ImageList(batch, [t.shape[-2:] for t in tensors])
