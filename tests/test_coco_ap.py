"""Box-AP restatement (counting_detr_amd/coco_ap.py; SURVEY.md 8f row 2, optional): hand-derived known answers and the invariants of the
COCOeval definition.  There is no pycocotools in this image and the reference holds no AP fixtures: PARITY UNPINNED (see the module header)."""
import json

import numpy as np
import pytest

from counting_detr_amd import coco_ap as ca


def _gt(b, **kw):
    d = {"bbox": [float(v) for v in b], "area": float(b[2] * b[3])}
    d.update(kw)
    return d


def _dt(b, s):
    return {"bbox": [float(v) for v in b], "score": float(s), "area": float(b[2] * b[3])}


def test_iou_matrix():
    iou = ca.box_iou_xywh([[0, 0, 10, 10], [5, 0, 10, 10], [100, 100, 1, 1]], [[0, 0, 10, 10], [0, 0, 20, 20]])
    np.testing.assert_allclose(iou, [[1.0, 0.25], [50 / 150, 100 / 400], [0.0, 0.0]])


def test_perfect_predictions_score_100():
    gts = {1: [_gt([10, 10, 20, 20]), _gt([100, 100, 50, 60])], 2: [_gt([5, 5, 120, 120])]}
    dts = {k: [_dt(g["bbox"], 0.9 - 0.1 * i) for i, g in enumerate(v)] for k, v in gts.items()}
    r = ca.summarize(gts, dts)
    assert r["AP"] == pytest.approx(100.0) and r["AP50"] == pytest.approx(100.0) and r["AP75"] == pytest.approx(100.0)
    assert r["APs"] == pytest.approx(100.0) and r["APm"] == pytest.approx(100.0) and r["APl"] == pytest.approx(100.0)   # 400 / 3000 / 14400 px^2


def test_hand_derived_case():
    """One image, two ground truths; detections by score: exact hit, a false positive, a hit with IoU 0.62.
    t in {0.50, 0.55, 0.60}: tp = 1,0,1 -> precision envelope 1, 2/3, 2/3 at recall .5, .5, 1 -> (51 + 50 * 2/3) / 101;
    t in {0.65 .. 0.95}:     tp = 1,0,0 -> recall stops at .5 -> 51 / 101."""
    s = 50 * 0.38 / 1.62                                   # x shift of a 50-wide box for IoU (50 - s) / (50 + s) = 0.62
    gts = {7: [_gt([10, 10, 40, 40]), _gt([100, 100, 50, 50])]}
    dts = {7: [_dt([10, 10, 40, 40], 0.9), _dt([300, 300, 30, 30], 0.8), _dt([100 + s, 100, 50, 50], 0.7)]}
    assert ca.box_iou_xywh([dts[7][2]["bbox"]], [gts[7][1]["bbox"]])[0, 0] == pytest.approx(0.62)
    lo, hi = (51 + 50 * 2 / 3) / 101, 51 / 101
    r = ca.summarize(gts, dts)
    assert r["AP50"] == pytest.approx(100 * lo, abs=1e-9)
    assert r["AP75"] == pytest.approx(100 * hi, abs=1e-9)
    assert r["AP"] == pytest.approx(100 * (3 * lo + 7 * hi) / 10, abs=1e-9)
    p = ca.average_precision(gts, dts)
    assert p.shape == (10, 101) and p[0, 50] == pytest.approx(1.0) and p[0, 51] == pytest.approx(2 / 3) and p[9, 51] == 0.0      # (tp / (tp + fp + eps))


def test_each_ground_truth_is_matched_once_and_best_iou_wins():
    gts = {1: [_gt([0, 0, 10, 10])]}
    dts = {1: [_dt([0, 0, 10, 10], 0.5), _dt([0, 0, 10, 10], 0.9)]}          # duplicate: the second (lower score) is a false positive
    p = ca.average_precision(gts, dts)
    assert np.allclose(p[:, 0], 1.0) and ca.summarize(gts, dts)["AP"] == pytest.approx(100.0)     # recall 1 reached at precision 1 by the first
    dts = {1: [_dt([0, 0, 10, 10], 0.5), _dt([200, 0, 10, 10], 0.9)]}        # the confident one misses: precision 1/2 at recall 1
    assert ca.summarize(gts, dts)["AP"] == pytest.approx(50.0)
    gts = {1: [_gt([0, 0, 10, 10]), _gt([2, 0, 10, 10])]}                    # one detection, two candidates: it takes the better IoU
    dts = {1: [_dt([2, 0, 10, 10], 0.9)]}
    _, m, _, n = ca._evaluate_image(dts[1], gts[1], ca.AREA_RNG["all"], 100)
    assert n == 2 and m[:, 0].all()


def test_ignored_ground_truth_and_area_ranges():
    gts = {1: [_gt([0, 0, 10, 10]), _gt([50, 50, 10, 10], iscrowd=1)]}
    dts = {1: [_dt([0, 0, 10, 10], 0.9), _dt([50, 50, 10, 10], 0.8)]}        # the second matches an ignored region: neither tp nor fp
    assert ca.summarize(gts, dts)["AP"] == pytest.approx(100.0)
    gts = {1: [_gt([0, 0, 10, 10]), _gt([100, 100, 200, 200])]}              # small + large
    dts = {1: [_dt([0, 0, 10, 10], 0.9)]}
    r = ca.summarize(gts, dts)
    assert r["APs"] == pytest.approx(100.0) and r["APl"] == pytest.approx(0.0) and np.isnan(r["APm"])
    assert r["AP"] == pytest.approx(100 * 51 / 101)


def test_order_invariance_and_max_dets():
    rng = np.random.default_rng(0)
    gts, dts = {}, {}
    for img in range(6):
        g = [[float(rng.integers(0, 200)), float(rng.integers(0, 200)), float(rng.integers(8, 120)), float(rng.integers(8, 120))] for _ in range(5)]
        gts[img] = [_gt(b) for b in g]
        dts[img] = [_dt([b[0] + rng.normal(0, 4), b[1] + rng.normal(0, 4), b[2], b[3]], rng.random()) for b in g] + \
                   [_dt([float(rng.integers(0, 300)), float(rng.integers(0, 300)), 20.0, 20.0], rng.random()) for _ in range(4)]
    base = ca.summarize(gts, dts)
    assert 0 < base["AP"] < 100
    perm = list(rng.permutation(6))
    g2 = {i: gts[i][::-1] for i in perm}
    d2 = {i: [dts[i][j] for j in rng.permutation(len(dts[i]))] for i in perm}
    again = ca.summarize(g2, d2)                                             # scores are distinct: any input order gives the same answer
    for k in base:
        assert again[k] == pytest.approx(base[k], abs=1e-9) or (np.isnan(again[k]) and np.isnan(base[k]))
    top1 = ca.summarize(gts, dts, max_det=1)
    only_best = {i: [max(v, key=lambda d: d["score"])] for i, v in dts.items()}
    assert top1["AP"] == pytest.approx(ca.summarize(gts, only_best)["AP"], abs=1e-9)


def test_json_pair_in_the_reference_wire_format(tmp_path):
    """predictions_<split>.json carries [cx, cy, w, h]; the reference turns it into int xywh before COCOeval (A2/eval_all.py:165-169)."""
    assert ca.reference_box([50.5, 40.0, 21.0, 10.0]) == [40, 35, 21, 10]
    gt = {"images": [{"id": 3}, {"id": 4}], "categories": [{"id": 1, "name": "fg"}],
          "annotations": [{"id": 1, "image_id": 3, "category_id": 1, "bbox": [40, 35, 21, 10], "area": 210, "iscrowd": 0},
                          {"id": 2, "image_id": 4, "category_id": 1, "bbox": [0, 0, 30, 30], "area": 900, "iscrowd": 0}]}
    pred = {"images": [{"id": 3}, {"id": 4}], "categories": [{"id": 1, "name": "fg"}],
            "annotations": [{"id": 1, "image_id": 3, "category_id": 1, "bbox": [50.5, 40.0, 21.0, 10.0], "score": 0.9, "point": [50, 40]},
                            {"id": 2, "image_id": 4, "category_id": 1, "bbox": [15, 15, 30, 30], "score": 0.8, "point": [15, 15]}]}
    pj, gj = tmp_path / "p.json", tmp_path / "g.json"
    pj.write_text(json.dumps(pred)); gj.write_text(json.dumps(gt))
    r = ca.ap_from_json(str(pj), str(gj))
    assert r["AP"] == pytest.approx(100.0) and r["AP50"] == pytest.approx(100.0)
    r = ca.ap_from_json(str(pj), str(gj), image_ids=[3])
    assert r["AP"] == pytest.approx(100.0)
