"""Datasets and COCO box evaluation either side of the hot path, on the CPU.

Pinned against the reference (tests/golden/datasets.{npz,json}, oracle/make_golden.py gen_datasets): Sim10kDataset /
KittiDataset items, the BoxList operations COCODataset applies to json boxes, prepare_for_coco_detection.
COCOeval restates pycocotools (absent from the reference tree and this image): known-answer cases worked out by hand
below and a separately written single-threshold AP.
"""
import json
import os

import numpy as np
import pytest
import torch

from scan_amd import coco_eval, datasets


@pytest.fixture(scope="module")
def gold(gold_dir):
    return (np.load(os.path.join(gold_dir, "datasets.npz")), json.load(open(os.path.join(gold_dir, "datasets.json"))))


@pytest.mark.parametrize("k", [0, 1, 2])
def test_voc_car_datasets_match_reference(gold, tmp_path, k):
    arr, meta = gold
    case = meta["voc"][k]
    root = tmp_path / case["tag"]
    for d in ("Annotations", "JPEGImages", os.path.join("ImageSets", "Main")):
        os.makedirs(root / d)
    for iid, f in case["files"].items():
        (root / "Annotations" / (iid + ".xml")).write_text(f["xml"])
        (root / "JPEGImages" / (iid + "." + case["ext"])).write_bytes(bytes.fromhex(f["image_hex"]))
    (root / "ImageSets" / "Main" / "train.txt").write_text("".join(i + "\n" for i in case["ids"]))
    cls = getattr(datasets, case["cls"])
    ds = cls(str(root), "train", use_difficult=case["use_difficult"], transforms=None, device=None)
    assert len(ds) == len(case["ids"])
    for i in range(len(ds)):
        image, (boxes, labels), idx = ds[i]
        assert idx == i
        assert np.array_equal(image.numpy(), arr["%s_img_%d" % (case["tag"], i)])
        assert np.array_equal(boxes.numpy(), arr["%s_boxes_%d" % (case["tag"], i)])
        assert np.array_equal(labels.numpy(), arr["%s_labels_%d" % (case["tag"], i)])
        assert ds.get_img_info(i) == case["items"][i]["info"]
        assert [image.shape[1], image.shape[0]] == case["items"][i]["size"]


def test_coco_box_conversion_and_clip_match_reference(gold):
    arr, _ = gold
    xyxy = datasets.xywh_to_xyxy(torch.from_numpy(arr["coco_xywh"]))
    assert np.array_equal(xyxy.numpy(), arr["coco_xyxy"])
    clipped, keep = datasets.clip_to_image(xyxy, tuple(int(v) for v in arr["coco_size"]))
    assert np.array_equal(clipped.numpy(), arr["coco_clipped"])
    labels = torch.arange(len(xyxy)) % 8 + 1
    assert np.array_equal(labels[keep].numpy(), arr["coco_clipped_labels"])
    assert 0 < len(clipped) < len(xyxy)  # the fixture does drop boxes


def test_prepare_for_coco_detection_matches_reference(gold):
    _, meta = gold
    p = meta["prepare"]

    class Dataset:
        id_to_img_map = {int(k): v for k, v in p["id_to_img_map"].items()}
        contiguous_category_id_to_json_id = {int(k): v for k, v in p["cat_map"].items()}

        def get_img_info(self, i):
            return p["infos"][i]

    preds = [(torch.tensor(q["boxes"], dtype=torch.float32).reshape(-1, 4), torch.tensor(q["scores"], dtype=torch.float32),
              torch.tensor(q["labels"], dtype=torch.int64), tuple(q["size"])) for q in p["predictions"]]
    res = datasets.prepare_for_coco_detection(preds, Dataset())
    assert res == p["results"]  # ids, categories, float boxes and scores: equal, not close


def _coco_json(images, anns, cats):
    return {"images": [{"id": i, "width": w, "height": h, "file_name": "%d.png" % i} for i, w, h in images],
            "annotations": [{"id": k + 1, "image_id": i, "category_id": c, "bbox": list(b), "area": b[2] * b[3],
                             "iscrowd": crowd} for k, (i, c, b, crowd) in enumerate(anns)],
            "categories": [{"id": c, "name": str(c)} for c in cats]}


def test_coco_dataset_indexing_rules():
    # categories in FILE order (33 before 24: pycocotools getCatIds keeps it), image ids sorted, crowd boxes dropped,
    # images whose boxes all have a side <= 1 or that have no annotation removed on request
    js = _coco_json([(7, 100, 50), (3, 80, 60), (9, 64, 64), (5, 32, 32)],
                    [(7, 24, (10, 10, 20, 20), 0), (7, 33, (0, 0, 200, 10), 0), (7, 24, (5, 5, 9, 9), 1),
                     (3, 33, (4, 4, 1, 30), 0), (9, 24, (60, 60, 10, 10), 0)], [33, 24])
    ds = datasets.COCODataset(datasets.CocoIndex(js), "/nonexistent", True, device=None)
    assert ds.ids == [7, 9]
    assert ds.json_category_id_to_contiguous_id == {33: 1, 24: 2}
    boxes, labels = ds.annotations(0)
    assert boxes.tolist() == [[10, 10, 29, 29], [0, 0, 99, 9]] and labels.tolist() == [2, 1]
    assert ds.annotations(1)[0].tolist() == [[60, 60, 63, 63]]
    assert ds.get_img_info(1)["id"] == 9 and ds.id_to_img_map == {0: 7, 1: 9}
    full = datasets.COCODataset(datasets.CocoIndex(js), "/nonexistent", False, device=None)
    assert full.ids == [3, 5, 7, 9] and full.annotations(1)[0].shape == (0, 4)
    assert full.annotations(0)[0].shape == (0, 4)  # x2 = x + max(w - 1, 0) = x: empty after conversion


def _evaluate(js, dets):
    gt = datasets.CocoIndex(js)
    ev = coco_eval.COCOeval(gt, gt.loadRes(dets) if dets else datasets.CocoIndex(), "bbox")
    ev.evaluate()
    ev.accumulate()
    return ev.summarize()


def test_cocoeval_perfect_detections():
    js = _coco_json([(1, 200, 200), (2, 200, 200)],
                    [(1, 1, (10, 10, 20, 20), 0), (1, 2, (50, 50, 40, 40), 0), (2, 1, (5, 5, 120, 120), 0)], [1, 2])
    dets = [{"image_id": a["image_id"], "category_id": a["category_id"], "bbox": a["bbox"], "score": 0.9 - 0.1 * k}
            for k, a in enumerate(js["annotations"])]
    s = _evaluate(js, dets)
    assert np.allclose(s[:3], 1.0) and np.allclose(s[6:9], 1.0)
    assert np.allclose(s[3:6], 1.0, atol=1e-12)  # 20x20 small, 40x40 medium, 120x120 large


def test_cocoeval_hand_computed_case():
    # one image, one category, two ground-truth boxes.  Detections in score order: d1 = gt1 exactly (IoU 1), d2 a false
    # positive, d3 covers gt2 with IoU 0.625 (50x50 vs 50x80 sharing x, y: 2500 / 4000).
    js = _coco_json([(1, 400, 400)], [(1, 1, (10, 10, 50, 50), 0), (1, 1, (200, 200, 50, 50), 0)], [1])
    dets = [{"image_id": 1, "category_id": 1, "bbox": [10, 10, 50, 50], "score": 0.9},
            {"image_id": 1, "category_id": 1, "bbox": [300, 20, 40, 40], "score": 0.8},
            {"image_id": 1, "category_id": 1, "bbox": [200, 200, 50, 80], "score": 0.7}]
    s = _evaluate(js, dets)
    # IoU thresholds 0.5, 0.55, 0.6 accept d3: (tp, fp) = (1,0), (1,1), (2,1); recall 0.5, 0.5, 1; precision made
    # monotone 1, 2/3, 2/3 -> 51 recall points (0 .. 0.5) at 1 and 50 (0.51 .. 1) at 2/3
    hi = (51 * 1.0 + 50 * (2.0 / 3.0)) / 101
    # thresholds >= 0.65 reject it: recall stops at 0.5 -> 51 points at 1, the rest 0
    lo = 51.0 / 101
    assert s[1] == pytest.approx(hi, abs=1e-12)
    assert s[2] == pytest.approx(lo, abs=1e-12)
    assert s[0] == pytest.approx((3 * hi + 7 * lo) / 10, abs=1e-12)
    assert s[4] == pytest.approx(s[0], abs=1e-12) and s[3] == -1 and s[5] == -1  # both boxes are "medium" (2500 px)
    assert s[6] == pytest.approx(0.5) and s[8] == pytest.approx((3 * 1.0 + 7 * 0.5) / 10)  # AR@1, AR@100


def test_cocoeval_crowd_and_area_rules():
    # a crowd region swallows any number of detections without making them false positives; an unmatched detection
    # outside the area range is ignored in that range
    js = _coco_json([(1, 500, 500)], [(1, 1, (0, 0, 300, 300), 1), (1, 1, (400, 400, 20, 20), 0)], [1])
    dets = [{"image_id": 1, "category_id": 1, "bbox": [10, 10, 50, 50], "score": 0.95},
            {"image_id": 1, "category_id": 1, "bbox": [100, 100, 60, 60], "score": 0.9},
            {"image_id": 1, "category_id": 1, "bbox": [400, 400, 20, 20], "score": 0.5}]
    s = _evaluate(js, dets)
    assert s[0] == pytest.approx(1.0) and s[1] == pytest.approx(1.0)   # the two high-score boxes inside the crowd do not count against precision
    assert s[3] == pytest.approx(1.0)    # small range: the 20x20 truth is found, the 50x50 / 60x60 boxes are out of range
    assert s[4] == -1 and s[5] == -1     # no medium / large truth that is not a crowd
    # without the crowd annotation the same detections are false positives ranked above the true one
    js2 = _coco_json([(1, 500, 500)], [(1, 1, (400, 400, 20, 20), 0)], [1])
    s2 = _evaluate(js2, dets)
    assert s2[1] == pytest.approx(1.0 / 3.0, abs=1e-12)


def test_cocoeval_no_detections_is_zero_not_undefined():
    js = _coco_json([(1, 100, 100)], [(1, 1, (10, 10, 40, 40), 0)], [1])
    s = _evaluate(js, [])
    assert s[0] == 0.0 and s[1] == 0.0 and s[8] == 0.0 and s[3] == -1


def _ap_single_threshold(js, dets, cat, thr):
    """separately written AP at one IoU threshold for one category without crowds: global score order, greedy match to
    the best free truth of the detection's image, interpolated precision at 101 recall points."""
    gts = {}
    for a in js["annotations"]:
        if a["category_id"] == cat:
            gts.setdefault(a["image_id"], []).append(a["bbox"])
    used = {i: [False] * len(b) for i, b in gts.items()}
    npos = sum(len(b) for b in gts.values())
    ds = sorted([d for d in dets if d["category_id"] == cat], key=lambda d: -d["score"])
    per_img = {}
    for d in ds:
        per_img.setdefault(d["image_id"], []).append(d)
    keep = set(id(d) for v in per_img.values() for d in v[:100])
    tp = []
    for d in ds:
        if id(d) not in keep:
            continue
        best, bi = thr, -1
        x, y, w, h = d["bbox"]
        for j, (gx, gy, gw, gh) in enumerate(gts.get(d["image_id"], [])):
            if used[d["image_id"]][j]:
                continue
            iw = min(x + w, gx + gw) - max(x, gx)
            ih = min(y + h, gy + gh) - max(y, gy)
            inter = max(iw, 0) * max(ih, 0)
            iou = inter / (w * h + gw * gh - inter)
            if iou >= best:
                best, bi = iou, j
        if bi >= 0:
            used[d["image_id"]][bi] = True
        tp.append(bi >= 0)
    tp = np.array(tp, bool)
    ctp, cfp = np.cumsum(tp), np.cumsum(~tp)
    rec = ctp / npos
    prec = ctp / np.maximum(ctp + cfp, 1e-300)
    env = np.maximum.accumulate(prec[::-1])[::-1] if len(prec) else prec
    out = []
    for r in np.linspace(0, 1, 101):
        k = np.searchsorted(rec, r, side="left")
        out.append(env[k] if k < len(env) else 0.0)
    return float(np.mean(out))


def test_cocoeval_ap50_against_separate_implementation():
    rng = np.random.RandomState(5)
    images = [(i + 1, 300, 300) for i in range(6)]
    anns, dets = [], []
    for i, _, _ in images:
        for _ in range(rng.randint(1, 6)):
            x, y, w, h = rng.randint(0, 200), rng.randint(0, 200), rng.randint(8, 90), rng.randint(8, 90)
            c = int(rng.randint(1, 3))
            anns.append((i, c, (x, y, w, h), 0))
            if rng.rand() < 0.8:  # a jittered detection of it
                j = rng.randint(-6, 7, 4)
                dets.append({"image_id": i, "category_id": c, "score": float(rng.rand()),
                             "bbox": [float(x + j[0]), float(y + j[1]), float(max(w + j[2], 2)), float(max(h + j[3], 2))]})
        for _ in range(rng.randint(0, 4)):  # clutter
            dets.append({"image_id": i, "category_id": int(rng.randint(1, 3)), "score": float(rng.rand()),
                         "bbox": [float(rng.randint(0, 250)), float(rng.randint(0, 250)), 30.0, 30.0]})
    js = _coco_json(images, anns, [1, 2])
    gt = datasets.CocoIndex(js)
    ev = coco_eval.COCOeval(gt, gt.loadRes(dets), "bbox")
    ev.evaluate()
    ev.accumulate()
    s = ev.summarize()
    for thr, idx in ((0.5, 1), (0.75, 2)):
        ours = [float(np.mean(ev.eval["precision"][list(ev.params.iouThrs).index(t), :, k, 0, 2]))
                for k in range(2) for t in ev.params.iouThrs if abs(t - thr) < 1e-9]
        ref = [_ap_single_threshold(js, dets, c, thr) for c in (1, 2)]
        assert ours == pytest.approx(ref, abs=1e-12)
        assert s[idx] == pytest.approx(np.mean(ref), abs=1e-12)
    assert 0.2 < s[1] < 1.0


def test_validation_writes_results_and_gate(tmp_path):
    js = _coco_json([(4, 200, 100), (2, 200, 100)], [(4, 24, (10, 10, 50, 50), 0), (2, 25, (20, 20, 30, 30), 0)], [24, 25])
    ds = datasets.COCODataset(datasets.CocoIndex(js), "/nonexistent", True, device=None)
    # detections in a 400x200 (2x) frame: they come back halved and as xywh, labels as json ids
    preds = [(torch.tensor([[40.0, 40.0, 78.0, 78.0]]), torch.tensor([0.9]), torch.tensor([2]), (400, 200)),
             (torch.tensor([[20.0, 20.0, 118.0, 118.0]]), torch.tensor([0.8]), torch.tensor([1]), (400, 200))]
    results, raw = coco_eval.do_coco_validation(ds, preds, str(tmp_path))
    assert raw["bbox"] == [{"image_id": 2, "category_id": 25, "bbox": [20.0, 20.0, 20.0, 20.0], "score": pytest.approx(0.9)},
                           {"image_id": 4, "category_id": 24, "bbox": [10.0, 10.0, 50.0, 50.0], "score": pytest.approx(0.8)}]
    assert json.load(open(tmp_path / "bbox.json")) == raw["bbox"]
    assert results.results["bbox"]["AP50"] == pytest.approx((1.0 + 0.0) / 2)  # 25: IoU 400/900 < 0.5; 24: exact
    gate = coco_eval.TargetGate(initial_ap50=30, val_type="AP50", val_iter=100)
    assert not gate.forward_target and gate.due(200) and not gate.due(150)
    assert gate.update(results) and gate.forward_target and gate.ap50_emp == pytest.approx(50.0)
    assert not gate.update(results)  # not a new best
    low = coco_eval.TargetGate(initial_ap50=60)
    assert not low.update(results) and not low.forward_target  # below the initial bar: no switch, no checkpoint


def test_ap50_against_reference_voc_evaluator(gold_dir):
    """AP50 cross-checked against REFERENCE-HELD code: tests/golden/voc_ap50.json holds a synthetic detection set and what the
    reference's pure-numpy VOC evaluator (data/datasets/evaluation/voc/voc_eval.py:48-200, use_07_metric=False) makes of it
    (oracle/make_golden.py gen_voc_ap: per-class AP, and the precision / recall arrays behind it).  The set is built so that the
    VOC and COCO matching rules coincide (truth of a class disjoint within an image, no difficult / crowd truth, distinct
    scores, <= 100 detections per image, integer boxes handed to each evaluator in its own pixel convention for the same
    geometric boxes -- see gen_voc_ap's docstring).  Then
      (1) the true / false-positive sequence COCOeval derives at IoU 0.5 must reproduce the reference's precision / recall
          arrays element by element (matching, score order and counting pinned exactly), so that
      (2) its AP50 = the 101-point sample of the reference's own precision-recall curve (formed here from the stored arrays,
          1e-12), and
      (3) lies within the interpolation error of the reference's exact-area AP (<= 1 / 101 + the recall granularity).
    AP at the other IoU thresholds, the area ranges and crowd handling remain parity-unpinned (no reference-held code)."""
    g = json.load(open(os.path.join(gold_dir, "voc_ap50.json")))
    W, H = g["size"]
    cats = list(range(1, g["n_classes"] + 1))
    js = _coco_json([(im["id"], W, H) for im in g["images"]],
                    [(im["id"], c, (x, y, w, h), 0) for im in g["images"] for (c, x, y, w, h) in im["gt"]], cats)
    dets = [{"image_id": im["id"], "category_id": c, "bbox": [x, y, w, h], "score": s}
            for im in g["images"] for (c, x, y, w, h, s) in im["dt"]]
    gt = datasets.CocoIndex(js)
    ev = coco_eval.COCOeval(gt, gt.loadRes(dets), "bbox")
    ev.evaluate()
    ev.accumulate()
    stats = ev.summarize()
    p = ev.params
    t50 = int(np.where(np.isclose(p.iouThrs, 0.5))[0][0])
    a_all, m100, I = p.areaRngLbl.index("all"), p.maxDets.index(100), len(p.imgIds)
    aps = []
    for k, c in enumerate(p.catIds):
        ref_prec, ref_rec = g["voc_prec"][c], g["voc_rec"][c]
        assert ref_prec is not None and ref_rec is not None
        # (1) COCOeval's per-image matches at IoU 0.5, in global score order
        E = [e for e in (ev.evalImgs[k * len(p.areaRng) * I + a_all * I + i] for i in range(I)) if e is not None]
        sc = np.concatenate([e["dtScores"] for e in E])
        order = np.argsort(-sc, kind="mergesort")
        tp = np.concatenate([e["dtMatches"][t50] for e in E])[order] > 0
        assert not np.concatenate([e["dtIgnore"][t50] for e in E]).any()
        ctp, cfp = np.cumsum(tp), np.cumsum(~tp)
        n_pos = sum(1 for im in g["images"] for r in im["gt"] if r[0] == c)
        np.testing.assert_allclose(ctp / (ctp + cfp), np.asarray(ref_prec), rtol=0, atol=1e-15)
        np.testing.assert_allclose(ctp / n_pos, np.asarray(ref_rec), rtol=0, atol=1e-15)
        # (2) the 101-point sample of the reference's curve
        env = np.maximum.accumulate(np.asarray(ref_prec)[::-1])[::-1]
        idx = np.searchsorted(np.asarray(ref_rec), p.recThrs, side="left")
        sampled = float(np.mean([env[j] if j < len(env) else 0.0 for j in idx]))
        mine = float(np.mean(ev.eval["precision"][t50, :, k, a_all, m100]))
        assert mine == pytest.approx(sampled, abs=1e-12), (c, mine, sampled)
        # (3) the reference's own AP (exact area under the monotone curve)
        assert abs(mine - g["voc_ap"][c]) <= 1.0 / 101 + 1.0 / n_pos, (c, mine, g["voc_ap"][c])
        aps.append(mine)
    assert stats[1] == pytest.approx(float(np.mean(aps)), abs=1e-12)
    assert abs(stats[1] - g["voc_map"]) <= 0.012, (stats[1], g["voc_map"])
    print("AP50 per class: COCOeval %s | reference VOC evaluator %s" % ([round(a, 4) for a in aps],
                                                                       [round(g["voc_ap"][c], 4) for c in cats]))
