"""COCO box evaluation for the in-loop validation that gates ``forward_target`` (reference engine/trainer.py:350,
465-479 -> engine/inference.py -> data/datasets/evaluation/coco/coco_eval.py:13-67,305-328,364-404,453-485).

The reference delegates the arithmetic to pycocotools.cocoeval.COCOeval, a third-party package that is neither in
/root/reference nor installed in this image and is not version-pinned by the reference (INSTALL.md: "pip install
pycocotools" / cocoapi master).  ``COCOeval`` below restates its published bbox algorithm (cocoapi
PythonAPI/pycocotools/cocoeval.py: _prepare, computeIoU, evaluateImg, accumulate, summarize; common/maskApi.c bbIou):

  * ground truth is ignored when it is a crowd or its ``area`` lies outside the area range; detections are taken in
    descending score order (stable), at most maxDet per image and category
  * a detection is matched greedily, per IoU threshold, to the not-yet-matched (or crowd) ground truth of highest
    IoU >= threshold, real ground truth before ignored ones, the later one on ties; IoU = inter / union with
    exclusive box extents (x + w), union = detection area for crowds
  * unmatched detections whose area lies outside the range are ignored
  * precision is made monotone from the right and sampled at 101 recall thresholds with searchsorted(side="left")
  * stats = AP, AP50, AP75, APs, APm, APl, AR@1, AR@10, AR@100, ARs, ARm, ARl; -1 where nothing is defined

Parity: pycocotools itself is absent, so AP@[.5:.95], AP75, the area ranges, AR and crowd handling are PARITY-UNPINNED
(known-answer cases worked out by hand and an independent single-threshold AP in tests/test_datasets_eval.py).  AP50 -- the
one number the training loop consumes (VAL_TYPE AP50 gates forward_target, trainer.py:350,465-479) -- IS cross-checked against
reference-held code: on a set where the VOC and COCO matching rules coincide, the true / false-positive sequence at IoU 0.5
reproduces the precision / recall arrays of the reference's numpy evaluator (data/datasets/evaluation/voc/voc_eval.py:48-200)
element by element, and AP50 equals the 101-point sample of that curve (test_ap50_against_reference_voc_evaluator,
tests/golden/voc_ap50.json written by oracle/make_golden.py gen_voc_ap).
"""
import json
import tempfile
from collections import OrderedDict, defaultdict

import numpy as np

from .datasets import CocoIndex, prepare_for_coco_detection


class Params:
    def __init__(self):
        self.imgIds = []
        self.catIds = []
        self.iouThrs = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
        self.recThrs = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
        self.maxDets = [1, 10, 100]
        self.areaRng = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
        self.areaRngLbl = ["all", "small", "medium", "large"]
        self.useCats = 1
        self.iouType = "bbox"


def box_iou_xywh(dt, gt, iscrowd):
    """maskApi.c bbIou: dt [D, 4], gt [G, 4] xywh, iscrowd [G] -> [D, G] float64."""
    dt = np.asarray(dt, np.float64).reshape(-1, 4)
    gt = np.asarray(gt, np.float64).reshape(-1, 4)
    crowd = np.asarray(iscrowd, bool).reshape(-1)
    da = dt[:, 2] * dt[:, 3]
    ga = gt[:, 2] * gt[:, 3]
    w = np.minimum(dt[:, None, 0] + dt[:, None, 2], gt[None, :, 0] + gt[None, :, 2]) - np.maximum(dt[:, None, 0], gt[None, :, 0])
    h = np.minimum(dt[:, None, 1] + dt[:, None, 3], gt[None, :, 1] + gt[None, :, 3]) - np.maximum(dt[:, None, 1], gt[None, :, 1])
    inter = np.clip(w, 0, None) * np.clip(h, 0, None)
    union = np.where(crowd[None, :], da[:, None], da[:, None] + ga[None, :] - inter)
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / union


class COCOeval:
    def __init__(self, coco_gt, coco_dt, iou_type="bbox"):
        assert iou_type == "bbox", "only box evaluation is on the SCAN path (iou_types = ('bbox',), trainer.py:104)"
        self.cocoGt, self.cocoDt = coco_gt, coco_dt
        self.params = Params()
        self.params.imgIds = sorted(coco_gt.getImgIds())
        self.params.catIds = sorted(coco_gt.getCatIds())
        self.evalImgs, self.eval, self.stats = [], {}, []

    def _prepare(self):
        p = self.params
        img_set, cat_set = set(p.imgIds), set(p.catIds)
        self._gts, self._dts = defaultdict(list), defaultdict(list)
        for a in self.cocoGt.dataset.get("annotations", []):
            if a["image_id"] in img_set and a["category_id"] in cat_set:
                g = dict(a)
                g["ignore"] = bool(g.get("iscrowd", 0))
                self._gts[g["image_id"], g["category_id"]].append(g)
        for a in self.cocoDt.dataset.get("annotations", []):
            if a["image_id"] in img_set and a["category_id"] in cat_set:
                self._dts[a["image_id"], a["category_id"]].append(a)

    def evaluate(self):
        p = self.params
        p.imgIds = list(np.unique(p.imgIds))
        p.catIds = list(np.unique(p.catIds))
        p.maxDets = sorted(p.maxDets)
        self._prepare()
        self.ious = {(i, c): self._compute_iou(i, c) for i in p.imgIds for c in p.catIds}
        max_det = p.maxDets[-1]
        self.evalImgs = [self._evaluate_img(i, c, rng, max_det) for c in p.catIds for rng in p.areaRng for i in p.imgIds]

    def _compute_iou(self, img, cat):
        gt, dt = self._gts[img, cat], self._dts[img, cat]
        if len(gt) == 0 and len(dt) == 0:
            return []
        order = np.argsort([-d["score"] for d in dt], kind="mergesort")
        dt = [dt[i] for i in order][:self.params.maxDets[-1]]
        return box_iou_xywh([d["bbox"] for d in dt], [g["bbox"] for g in gt], [int(g.get("iscrowd", 0)) for g in gt])

    def _evaluate_img(self, img, cat, rng, max_det):
        p = self.params
        gt, dt = self._gts[img, cat], self._dts[img, cat]
        if len(gt) == 0 and len(dt) == 0:
            return None
        g_ign = np.array([1 if (g["ignore"] or g["area"] < rng[0] or g["area"] > rng[1]) else 0 for g in gt], int)
        gtind = np.argsort(g_ign, kind="mergesort")
        gt = [gt[i] for i in gtind]
        dtind = np.argsort([-d["score"] for d in dt], kind="mergesort")
        dt = [dt[i] for i in dtind[:max_det]]
        iscrowd = [int(g.get("iscrowd", 0)) for g in gt]
        ious = self.ious[img, cat][:, gtind] if len(self.ious[img, cat]) > 0 else self.ious[img, cat]
        T, G, D = len(p.iouThrs), len(gt), len(dt)
        gtm, dtm = np.zeros((T, G)), np.zeros((T, D))
        gt_ig = g_ign[gtind]
        dt_ig = np.zeros((T, D))
        if len(ious) != 0:
            for ti, t in enumerate(p.iouThrs):
                for di, d in enumerate(dt):
                    best, m = min(t, 1 - 1e-10), -1
                    for gi in range(G):
                        if gtm[ti, gi] > 0 and not iscrowd[gi]:
                            continue
                        if m > -1 and gt_ig[m] == 0 and gt_ig[gi] == 1:
                            break
                        if ious[di, gi] < best:
                            continue
                        best, m = ious[di, gi], gi
                    if m == -1:
                        continue
                    dt_ig[ti, di] = gt_ig[m]
                    dtm[ti, di] = gt[m]["id"]
                    gtm[ti, m] = d["id"]
        out_of_range = np.array([d["area"] < rng[0] or d["area"] > rng[1] for d in dt]).reshape((1, D))
        dt_ig = np.logical_or(dt_ig, np.logical_and(dtm == 0, np.repeat(out_of_range, T, 0)))
        return {"image_id": img, "category_id": cat, "aRng": rng, "maxDet": max_det,
                "dtMatches": dtm, "gtMatches": gtm, "dtScores": [d["score"] for d in dt], "gtIgnore": gt_ig,
                "dtIgnore": dt_ig}

    def accumulate(self):
        p = self.params
        T, R, K, A, M = len(p.iouThrs), len(p.recThrs), len(p.catIds), len(p.areaRng), len(p.maxDets)
        precision = -np.ones((T, R, K, A, M))
        recall = -np.ones((T, K, A, M))
        scores = -np.ones((T, R, K, A, M))
        I = len(p.imgIds)
        for k in range(K):
            for a in range(A):
                for m, max_det in enumerate(p.maxDets):
                    E = [self.evalImgs[k * A * I + a * I + i] for i in range(I)]
                    E = [e for e in E if e is not None]
                    if len(E) == 0:
                        continue
                    dt_scores = np.concatenate([e["dtScores"][0:max_det] for e in E])
                    inds = np.argsort(-dt_scores, kind="mergesort")
                    dt_sorted = dt_scores[inds]
                    dtm = np.concatenate([e["dtMatches"][:, 0:max_det] for e in E], axis=1)[:, inds]
                    dt_ig = np.concatenate([e["dtIgnore"][:, 0:max_det] for e in E], axis=1)[:, inds]
                    gt_ig = np.concatenate([e["gtIgnore"] for e in E])
                    npig = np.count_nonzero(gt_ig == 0)
                    if npig == 0:
                        continue
                    tps = np.logical_and(dtm, np.logical_not(dt_ig))
                    fps = np.logical_and(np.logical_not(dtm), np.logical_not(dt_ig))
                    tp_sum = np.cumsum(tps, axis=1).astype(float)
                    fp_sum = np.cumsum(fps, axis=1).astype(float)
                    for t, (tp, fp) in enumerate(zip(tp_sum, fp_sum)):
                        nd = len(tp)
                        rc = tp / npig
                        pr = tp / (fp + tp + np.spacing(1))
                        q, ss = np.zeros((R,)), np.zeros((R,))
                        recall[t, k, a, m] = rc[-1] if nd else 0
                        pr = pr.tolist()
                        for i in range(nd - 1, 0, -1):
                            if pr[i] > pr[i - 1]:
                                pr[i - 1] = pr[i]
                        where = np.searchsorted(rc, p.recThrs, side="left")
                        for ri, pi in enumerate(where):
                            if pi >= nd:
                                break
                            q[ri] = pr[pi]
                            ss[ri] = dt_sorted[pi]
                        precision[t, :, k, a, m] = q
                        scores[t, :, k, a, m] = ss
        self.eval = {"params": p, "counts": [T, R, K, A, M], "precision": precision, "recall": recall, "scores": scores}

    def _summarize(self, ap, iou_thr=None, area="all", max_dets=100):
        p = self.params
        a = p.areaRngLbl.index(area)
        m = p.maxDets.index(max_dets)
        s = self.eval["precision"] if ap else self.eval["recall"]
        if iou_thr is not None:
            s = s[np.where(iou_thr == p.iouThrs)[0]]
        s = s[:, :, :, a, m] if ap else s[:, :, a, m]
        return -1 if len(s[s > -1]) == 0 else float(np.mean(s[s > -1]))

    def summarize(self):
        md = self.params.maxDets
        self.stats = np.array([
            self._summarize(1), self._summarize(1, iou_thr=.5, max_dets=md[2]), self._summarize(1, iou_thr=.75, max_dets=md[2]),
            self._summarize(1, area="small", max_dets=md[2]), self._summarize(1, area="medium", max_dets=md[2]),
            self._summarize(1, area="large", max_dets=md[2]),
            self._summarize(0, max_dets=md[0]), self._summarize(0, max_dets=md[1]), self._summarize(0, max_dets=md[2]),
            self._summarize(0, area="small", max_dets=md[2]), self._summarize(0, area="medium", max_dets=md[2]),
            self._summarize(0, area="large", max_dets=md[2])])
        return self.stats


class COCOResults:
    """reference coco_eval.py:364-404: results[iou_type][metric], -1 until updated."""
    METRICS = {"bbox": ["AP", "AP50", "AP75", "APs", "APm", "APl"]}

    def __init__(self, *iou_types):
        assert all(t in self.METRICS for t in iou_types)
        self.results = OrderedDict((t, OrderedDict((m, -1) for m in self.METRICS[t])) for t in iou_types)

    def update(self, coco_eval):
        if coco_eval is None:
            return
        res = self.results[coco_eval.params.iouType]
        for idx, metric in enumerate(self.METRICS[coco_eval.params.iouType]):
            res[metric] = coco_eval.stats[idx]

    def __repr__(self):
        return repr(self.results)


def evaluate_predictions_on_coco(coco_gt, coco_results, json_result_file, iou_type="bbox"):
    """reference coco_eval.py:305-328: the results go through their json file, as there."""
    with open(json_result_file, "w") as f:
        json.dump(coco_results, f)
    coco_dt = coco_gt.loadRes(str(json_result_file)) if coco_results else CocoIndex()
    ev = COCOeval(coco_gt, coco_dt, iou_type)
    ev.evaluate()
    ev.accumulate()
    ev.summarize()
    return ev


def do_coco_validation(dataset, predictions, output_folder=None, iou_types=("bbox",)):
    """reference do_coco_validation / do_coco_evaluation for FCOS (box_only False, no expected results):
    (COCOResults, {"bbox": result dicts})."""
    import os
    coco_results = {"bbox": prepare_for_coco_detection(predictions, dataset)}
    results = COCOResults(*iou_types)
    for iou_type in iou_types:
        with tempfile.NamedTemporaryFile() as f:
            path = os.path.join(output_folder, iou_type + ".json") if output_folder else f.name
            results.update(evaluate_predictions_on_coco(dataset.coco, coco_results[iou_type], path, iou_type))
    return results, coco_results


class TargetGate:
    """The dynamic switch of the target-domain graph branch (reference trainer.py:179-181,350,465-479): validation
    every VAL_ITER iterations sets AP50_emp = results['bbox'][VAL_TYPE] * 100; the next iterations run with
    forward_target = AP50_emp > SOLVER.INITIAL_AP50; a new best is reported so the caller can checkpoint."""

    def __init__(self, initial_ap50, val_type="AP50", val_iter=100, adapt_val_on=True):
        self.initial_ap50 = float(initial_ap50)
        self.val_type, self.val_iter, self.adapt_val_on = val_type, int(val_iter), bool(adapt_val_on)
        self.ap50_emp = 0.0              # trainer.py:150
        self.best = self.initial_ap50    # trainer.py:149: a checkpoint is written only above the initial bar

    @property
    def forward_target(self):
        return self.ap50_emp > self.initial_ap50

    def due(self, iteration):
        return self.adapt_val_on and iteration % self.val_iter == 0

    def update(self, results):
        """results: COCOResults of the validation run; returns True when it is a new best (checkpoint trigger)."""
        self.ap50_emp = results.results["bbox"][self.val_type] * 100
        if self.ap50_emp > self.best:
            self.best = self.ap50_emp
            return True
        return False
