"""Datasets either side of the hot path: the annotation formats the SCAN configs train and
validate on, read into the (boxes xyxy, labels) targets the path consumes, and detections written back out in
the COCO result format.

  COCODataset      fcos_core/data/datasets/coco.py:39-101 -- Cityscapes / Foggy Cityscapes in COCO json
                   (config/paths_catalog.py:101-124).  The reference leans on torchvision's CocoDetection and
                   pycocotools' COCO index; neither is a dependency here: ``CocoIndex`` is the part of that index the
                   path uses (image / annotation / category tables of the public COCO json layout).
  Sim10kDataset,   fcos_core/data/datasets/sim10k.py:17-119, kitti.py -- PASCAL-VOC xml, "car" only, 1-based pixel
  KittiDataset     indices made 0-based.
  prepare_for_coco_detection   fcos_core/data/datasets/evaluation/coco/coco_eval.py:69-98.

Box conventions are the reference BoxList's (structures/bounding_box.py:60-89,209-219): xywh -> xyxy with the
inclusive-pixel ``TO_REMOVE = 1``, clip to [0, size - 1], drop boxes that are empty after clipping.

Decoded frames are uploaded as ``data.U8Image`` (uint8 RGB on the device); JPEG / PNG decode is PIL on the host.
"""
import json
import os
import xml.etree.ElementTree as ET
from collections import defaultdict

import torch

from .data import U8Image

MIN_KEYPOINTS_PER_IMAGE = 10


class CocoIndex:
    """The slice of pycocotools.coco.COCO the path touches: ``imgs``, ``anns``, ``cats``, ``getCatIds`` (file
    order), ``getImgIds``, ``getAnnIds(imgIds=..)`` (file order within an image), ``loadAnns`` and ``loadRes`` for
    box results (area = w * h, ids 1..n, iscrowd 0)."""

    def __init__(self, annotation=None):
        if annotation is None:
            annotation = {"images": [], "annotations": [], "categories": []}
        elif not isinstance(annotation, dict):
            with open(annotation) as f:
                annotation = json.load(f)
        self.dataset = annotation
        self.imgs = {im["id"]: im for im in annotation.get("images", [])}
        self.anns = {a["id"]: a for a in annotation.get("annotations", [])}
        self.cats = {c["id"]: c for c in annotation.get("categories", [])}
        self.img_to_anns = defaultdict(list)
        for a in annotation.get("annotations", []):
            self.img_to_anns[a["image_id"]].append(a)

    def getCatIds(self):
        return [c["id"] for c in self.dataset.get("categories", [])]

    def getImgIds(self):
        return list(self.imgs.keys())

    def getAnnIds(self, imgIds, iscrowd=None):
        ids = imgIds if isinstance(imgIds, (list, tuple)) else [imgIds]
        anns = [a for i in ids for a in self.img_to_anns.get(i, [])]
        if iscrowd is not None:
            anns = [a for a in anns if a.get("iscrowd", 0) == iscrowd]
        return [a["id"] for a in anns]

    def loadAnns(self, ids):
        return [self.anns[i] for i in ids]

    def loadRes(self, results):
        """detections (list of {image_id, category_id, bbox xywh, score} or the json file holding it) as an index
        over the same images and categories."""
        if not isinstance(results, list):
            with open(results) as f:
                results = json.load(f)
        known = set(self.imgs.keys())
        assert all(r["image_id"] in known for r in results), "results do not correspond to the annotation file"
        anns = []
        for k, r in enumerate(results):
            a = dict(r)
            x, y, w, h = r["bbox"]
            a["area"] = w * h
            a["id"] = k + 1
            a["iscrowd"] = 0
            anns.append(a)
        return CocoIndex({"images": list(self.dataset.get("images", [])), "annotations": anns,
                          "categories": list(self.dataset.get("categories", []))})


def has_valid_annotation(anno):
    """reference coco.py:13-36: not empty, not all boxes with a side <= 1, >= 10 visible keypoints for keypoint sets."""
    if len(anno) == 0:
        return False
    if all(any(o <= 1 for o in obj["bbox"][2:]) for obj in anno):
        return False
    if "keypoints" not in anno[0]:
        return True
    visible = sum(sum(1 for v in ann["keypoints"][2::3] if v > 0) for ann in anno)
    return visible >= MIN_KEYPOINTS_PER_IMAGE


def xywh_to_xyxy(boxes):
    """BoxList(mode="xywh").convert("xyxy") (bounding_box.py:75-89): x2 = x + max(w - 1, 0)."""
    boxes = torch.as_tensor(boxes, dtype=torch.float32).reshape(-1, 4)
    x, y, w, h = boxes.unbind(1)
    return torch.stack([x, y, x + (w - 1).clamp(min=0), y + (h - 1).clamp(min=0)], 1)


def xyxy_to_xywh(boxes):
    """BoxList.convert("xywh") (bounding_box.py:66-71): w = x2 - x1 + 1."""
    x1, y1, x2, y2 = boxes.unbind(1)
    return torch.stack([x1, y1, x2 - x1 + 1, y2 - y1 + 1], 1)


def clip_to_image(boxes, size, remove_empty=True):
    """BoxList.clip_to_image (bounding_box.py:209-219) for size = (width, height): (clipped boxes, keep mask)."""
    w, h = size
    b = boxes.clone()
    b[:, 0].clamp_(min=0, max=w - 1)
    b[:, 1].clamp_(min=0, max=h - 1)
    b[:, 2].clamp_(min=0, max=w - 1)
    b[:, 3].clamp_(min=0, max=h - 1)
    keep = (b[:, 3] > b[:, 1]) & (b[:, 2] > b[:, 0]) if remove_empty else torch.ones(len(b), dtype=torch.bool)
    return b[keep], keep


def _open_rgb(path, device):
    """decode on the host (PIL), upload as a U8Image; device=None keeps the decoded uint8 [H, W, 3] tensor on the
    host -- for annotation tooling and tests; the transforms (HIP kernels) accept device images only."""
    import numpy as np
    from PIL import Image
    with Image.open(path) as im:
        arr = np.array(im.convert("RGB"), dtype=np.uint8)
    return torch.from_numpy(arr) if device is None else U8Image.from_numpy(arr, device=device)


class COCODataset:
    """reference data/datasets/coco.py:39-101.  ``self[i]`` -> (image, (boxes xyxy float32 [G, 4], labels int64 [G]),
    i); crowd annotations dropped, category ids mapped to 1..K in file order, boxes clipped / emptied ones removed,
    then ``transforms(image, target)``."""

    def __init__(self, ann_file, root, remove_images_without_annotations, transforms=None, device="cuda"):
        self.root = root
        self.coco = ann_file if isinstance(ann_file, CocoIndex) else CocoIndex(ann_file)
        self.ids = sorted(self.coco.imgs.keys())
        if remove_images_without_annotations:
            self.ids = [i for i in self.ids
                        if has_valid_annotation(self.coco.loadAnns(self.coco.getAnnIds(imgIds=i, iscrowd=None)))]
        self.json_category_id_to_contiguous_id = {v: i + 1 for i, v in enumerate(self.coco.getCatIds())}
        self.contiguous_category_id_to_json_id = {v: k for k, v in self.json_category_id_to_contiguous_id.items()}
        self.id_to_img_map = {k: v for k, v in enumerate(self.ids)}
        self.transforms = transforms
        self.device = device

    def __len__(self):
        return len(self.ids)

    def annotations(self, idx):
        """the target of item idx without touching the image file"""
        img_id = self.ids[idx]
        info = self.coco.imgs[img_id]
        anno = [o for o in self.coco.loadAnns(self.coco.getAnnIds(imgIds=img_id)) if o["iscrowd"] == 0]
        boxes = xywh_to_xyxy([o["bbox"] for o in anno])
        labels = torch.tensor([self.json_category_id_to_contiguous_id[o["category_id"]] for o in anno],
                              dtype=torch.int64)
        boxes, keep = clip_to_image(boxes, (info["width"], info["height"]), remove_empty=True)
        return boxes, labels[keep]

    def __getitem__(self, idx):
        info = self.coco.imgs[self.ids[idx]]
        image = _open_rgb(os.path.join(self.root, info["file_name"]), self.device)
        # the reference clips to the decoded image's size (img.size); the json's width / height describe that image
        target = self.annotations(idx)
        if self.transforms is not None:
            image, target = self.transforms(image, target)
        return image, target, idx

    def get_img_info(self, index):
        return self.coco.imgs[self.id_to_img_map[index]]


class _VocCarDataset:
    """reference sim10k.py / kitti.py: ImageSets/Main/<split>.txt, Annotations/<id>.xml, JPEGImages/<id>.<ext>; only
    objects named "car"; xmin..ymax are 1-based pixel indices."""
    CLASSES = ("__background__ ", "car")
    EXT = "jpg"
    READ_DIFFICULT = True

    def __init__(self, data_dir, split, use_difficult=False, transforms=None, device="cuda"):
        self.root = data_dir
        self.image_set = split
        self.keep_difficult = use_difficult
        self.transforms = transforms
        self.device = device
        self._annopath = os.path.join(self.root, "Annotations", "%s.xml")
        self._imgpath = os.path.join(self.root, "JPEGImages", "%s." + self.EXT)
        with open(os.path.join(self.root, "ImageSets", "Main", "%s.txt" % split)) as f:
            self.ids = [x.strip("\n") for x in f.readlines()]
        self.id_to_img_map = {k: v for k, v in enumerate(self.ids)}
        self.class_to_ind = dict(zip(self.CLASSES, range(len(self.CLASSES))))

    def __len__(self):
        return len(self.ids)

    def get_groundtruth(self, index):
        """(boxes, labels, difficult, (width, height)) before clipping (reference get_groundtruth)."""
        root = ET.parse(self._annopath % self.ids[index]).getroot()
        boxes, labels, difficult = [], [], []
        for obj in root.iter("object"):
            hard = int(obj.find("difficult").text) == 1 if self.READ_DIFFICULT else False
            if not self.keep_difficult and hard:
                continue
            name = obj.find("name").text.lower().strip()
            if name != "car":
                continue
            bb = obj.find("bndbox")
            boxes.append([int(bb.find(k).text) - 1 for k in ("xmin", "ymin", "xmax", "ymax")])
            labels.append(self.class_to_ind[name])
            difficult.append(hard)
        size = root.find("size")
        wh = (int(size.find("width").text), int(size.find("height").text))
        return (torch.tensor(boxes, dtype=torch.float32).reshape(-1, 4), torch.tensor(labels, dtype=torch.int64),
                torch.tensor(difficult, dtype=torch.bool), wh)

    def annotations(self, index):
        boxes, labels, _, wh = self.get_groundtruth(index)
        boxes, keep = clip_to_image(boxes, wh, remove_empty=True)
        return boxes, labels[keep]

    def __getitem__(self, index):
        image = _open_rgb(self._imgpath % self.ids[index], self.device)
        target = self.annotations(index)
        if self.transforms is not None:
            image, target = self.transforms(image, target)
        return image, target, index

    def get_img_info(self, index):
        size = ET.parse(self._annopath % self.ids[index]).getroot().find("size")
        return {"height": int(size.find("height").text), "width": int(size.find("width").text)}

    def map_class_id_to_class_name(self, class_id):
        return self.CLASSES[class_id]


class Sim10kDataset(_VocCarDataset):
    EXT = "jpg"
    READ_DIFFICULT = True


class KittiDataset(_VocCarDataset):
    """kitti.py: png frames, every object counted as not difficult."""
    EXT = "png"
    READ_DIFFICULT = False


def resize_detections(boxes, from_size, to_size):
    """BoxList.resize on xyxy detections (bounding_box.py:91-131), sizes = (width, height)."""
    rw, rh = (float(s) / float(o) for s, o in zip(to_size, from_size))
    if rw == rh:
        return boxes * rw
    return boxes * boxes.new_tensor([rw, rh, rw, rh])


def prepare_for_coco_detection(predictions, dataset):
    """reference coco_eval.py:69-98.  predictions[i] = (boxes xyxy [n, 4], scores [n], labels [n], (width, height) the
    boxes live in) for dataset item i, or None / n = 0.  Boxes go back to the original frame size, to xywh, labels to
    the json category ids."""
    out = []
    for image_id, pred in enumerate(predictions):
        if pred is None or len(pred[0]) == 0:
            continue
        boxes, scores, labels, size = pred
        info = dataset.get_img_info(image_id)
        b = resize_detections(boxes.detach().float().cpu(), size, (info["width"], info["height"]))
        b = xyxy_to_xywh(b).tolist()
        s = scores.detach().float().cpu().tolist()
        lab = [dataset.contiguous_category_id_to_json_id[int(i)] for i in labels.detach().cpu().tolist()]
        original_id = dataset.id_to_img_map[image_id]
        out.extend({"image_id": original_id, "category_id": lab[k], "bbox": box, "score": s[k]}
                   for k, box in enumerate(b))
    return out
