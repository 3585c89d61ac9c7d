"""Box AP of a prediction json against a COCO-style ground-truth json, without detectron2 / pycocotools (optional part).

The reference's offline evaluator (A2/eval_all.py:141-279, 285-312, 496-531) hands the predictions to pycocotools' `COCOeval`
(`iouType="bbox"`, `maxDets = [900, 1000, 1100]`, summary read at `maxDets[2]`) after turning each predicted `[cx, cy, w, h]` into
`[int(cx - w/2), int(cy - h/2), int(w), int(h)]` (A2/eval_all.py:165-169) and reports AP, AP50, AP75, APs, APm, APl x 100.
pycocotools is a third-party dependency that is neither vendored in the reference tree nor installed in this image (its version is
not pinned by the reference either), so this module restates its PUBLISHED algorithm (cocoeval.py: computeIoU / evaluateImg /
accumulate / summarize, bbox path, no crowd regions in FSC-147):

  * per image and category, detections by descending score (stable), at most maxDet of them; ground truths with `ignore` / `iscrowd`
    or an area outside the area range are "ignored" and sorted behind the others;
  * for each IoU threshold t in 0.50:0.05:0.95 a detection takes the still-unmatched ground truth of highest IoU >= min(t, 1 - 1e-10),
    preferring non-ignored ones (the scan stops at the first ignored ground truth once a non-ignored match is held); a detection
    matched to an ignored ground truth, or unmatched with an area outside the range, is ignored itself;
  * all images' detections merged by descending score (stable): cumulative tp / fp -> recall = tp / #non-ignored gt,
    precision = tp / (tp + fp + eps), made monotonically non-increasing from the right, sampled at the 101 recall thresholds 0:0.01:1
    with `searchsorted(recall, thr, side="left")` (0 beyond the reached recall);
  * AP = mean over thresholds and recall samples of the precision (entries of -1 = no ground truth are skipped).

PARITY UNPINNED: there is no pycocotools here to generate golden vectors from, and the reference holds no AP fixtures.  The tests
(`tests/test_coco_ap.py`) pin the restatement to hand-derived cases and to the invariants of the definition only.  It is host-side
post-processing on a json pair -- not part of the step `bench.py` measures.
"""
import json

import numpy as np

IOU_THRS = np.linspace(0.5, 0.95, int(np.round((0.95 - 0.5) / 0.05)) + 1, endpoint=True)
REC_THRS = np.linspace(0.0, 1.0, int(np.round((1.0 - 0.0) / 0.01)) + 1, endpoint=True)
AREA_RNG = {"all": (0.0, 1e5 ** 2), "small": (0.0, 32.0 ** 2), "medium": (32.0 ** 2, 96.0 ** 2), "large": (96.0 ** 2, 1e5 ** 2)}
MAX_DETS = 1100          # A2/eval_all.py:511-512: maxDets = [900, 1000, 1100], AP summarised at the last one


def reference_box(b):
    """[cx, cy, w, h] of the prediction json -> the [x, y, w, h] ints the reference feeds to COCOeval (A2/eval_all.py:165-169)."""
    cx, cy, w, h = b
    return [int(cx - w / 2), int(cy - h / 2), int(w), int(h)]


def box_iou_xywh(dt, gt):
    """IoU matrix [len(dt), len(gt)] of xywh boxes (pycocotools maskApi bbIou, no crowd)."""
    dt = np.asarray(dt, dtype=np.float64).reshape(-1, 4)
    gt = np.asarray(gt, dtype=np.float64).reshape(-1, 4)
    if len(dt) == 0 or len(gt) == 0:
        return np.zeros((len(dt), len(gt)))
    da, ga = dt[:, 2] * dt[:, 3], gt[:, 2] * gt[:, 3]
    w = np.minimum(dt[:, None, 0] + dt[:, None, 2], gt[None, :, 0] + gt[None, :, 2]) - np.maximum(dt[:, None, 0], gt[None, :, 0])
    h = np.minimum(dt[:, None, 1] + dt[:, None, 3], gt[None, :, 1] + gt[None, :, 3]) - np.maximum(dt[:, None, 1], gt[None, :, 1])
    inter = np.clip(w, 0, None) * np.clip(h, 0, None)
    union = da[:, None] + ga[None, :] - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = np.where(union > 0, inter / union, 0.0)
    return iou


def _evaluate_image(dts, gts, area_rng, max_det):
    """One (image, category, area range): -> (scores [D], matched [T, D] bool, det_ignored [T, D] bool, number of non-ignored gt)."""
    g_ig = np.array([bool(g.get("ignore", 0)) or bool(g.get("iscrowd", 0)) or g["area"] < area_rng[0] or g["area"] > area_rng[1] for g in gts],
                    dtype=bool)
    g_order = np.argsort(g_ig, kind="mergesort")                        # non-ignored first, original order otherwise
    gts = [gts[i] for i in g_order]
    g_ig = g_ig[g_order]
    d_order = np.argsort([-d["score"] for d in dts], kind="mergesort")[:max_det]
    dts = [dts[i] for i in d_order]
    ious = box_iou_xywh([d["bbox"] for d in dts], [g["bbox"] for g in gts])
    T, D, G = len(IOU_THRS), len(dts), len(gts)
    gtm = -np.ones((T, G), dtype=np.int64)
    dtm = -np.ones((T, D), dtype=np.int64)
    dt_ig = np.zeros((T, D), dtype=bool)
    for ti, t in enumerate(IOU_THRS):
        for di in range(D):
            best, m = min(t, 1 - 1e-10), -1
            for gi in range(G):
                if gtm[ti, gi] >= 0:                                    # taken (no crowd regions here)
                    continue
                if m > -1 and not g_ig[m] and g_ig[gi]:                 # holding a real match: ignored ones cannot replace it
                    break
                if ious[di, gi] < best:
                    continue
                best, m = ious[di, gi], gi
            if m == -1:
                continue
            dt_ig[ti, di] = g_ig[m]
            dtm[ti, di] = m
            gtm[ti, m] = di
    d_area = np.array([d.get("area", d["bbox"][2] * d["bbox"][3]) for d in dts], dtype=np.float64)
    out_of_range = (d_area < area_rng[0]) | (d_area > area_rng[1])
    dt_ig = dt_ig | ((dtm < 0) & out_of_range[None, :])
    return np.array([d["score"] for d in dts], dtype=np.float64), dtm >= 0, dt_ig, int((~g_ig).sum())


def average_precision(gt_by_img, dt_by_img, area="all", max_det=MAX_DETS):
    """`precision[T, R]` (COCOeval.eval["precision"][:, :, k, a, m]) for one category; -1 everywhere when there is no ground truth.
    gt_by_img / dt_by_img: {image_id: [ {bbox: xywh, area, (iscrowd), (ignore)} ]} / {image_id: [ {bbox: xywh, score} ]}."""
    rng = AREA_RNG[area]
    scores, matched, ignored, npig = [], [], [], 0
    for img in sorted(set(gt_by_img) | set(dt_by_img)):
        g, d = gt_by_img.get(img, []), dt_by_img.get(img, [])
        if not g and not d:
            continue
        s, m, ig, n = _evaluate_image(d, g, rng, max_det)
        scores.append(s); matched.append(m); ignored.append(ig)
        npig += n
    T, R = len(IOU_THRS), len(REC_THRS)
    precision = -np.ones((T, R))
    if npig == 0:
        return precision
    scores = np.concatenate(scores) if scores else np.zeros(0)
    order = np.argsort(-scores, kind="mergesort")
    matched = np.concatenate(matched, axis=1)[:, order] if matched else np.zeros((T, 0), dtype=bool)
    ignored = np.concatenate(ignored, axis=1)[:, order] if ignored else np.zeros((T, 0), dtype=bool)
    tps = np.cumsum(matched & ~ignored, axis=1, dtype=np.float64)
    fps = np.cumsum(~matched & ~ignored, axis=1, dtype=np.float64)
    for t in range(T):
        tp, fp = tps[t], fps[t]
        rc = tp / npig
        pr = tp / (fp + tp + np.spacing(1))
        q = np.zeros(R)
        pr = pr.tolist()
        for i in range(len(pr) - 1, 0, -1):                             # precision envelope
            if pr[i] > pr[i - 1]:
                pr[i - 1] = pr[i]
        inds = np.searchsorted(rc, REC_THRS, side="left")
        for ri, pi in enumerate(inds):
            if pi < len(pr):
                q[ri] = pr[pi]
        precision[t] = q
    return precision


def _mean(p):
    p = p[p > -1]
    return float(np.mean(p)) if p.size else -1.0


def summarize(gt_by_img, dt_by_img, max_det=MAX_DETS):
    """The six numbers of A2/eval_all.py:331 (x 100, NaN when undefined): AP, AP50, AP75, APs, APm, APl."""
    p_all = average_precision(gt_by_img, dt_by_img, "all", max_det)
    vals = {"AP": _mean(p_all), "AP50": _mean(p_all[np.isclose(IOU_THRS, 0.5)]), "AP75": _mean(p_all[np.isclose(IOU_THRS, 0.75)])}
    for key, area in (("APs", "small"), ("APm", "medium"), ("APl", "large")):
        vals[key] = _mean(average_precision(gt_by_img, dt_by_img, area, max_det))
    return {k: (v * 100 if v >= 0 else float("nan")) for k, v in vals.items()}


def ap_from_json(pred_json, gt_json, image_ids=None):
    """AP of `predictions_<split>.json` (the wire format of infer.py / A2/infer.py:84-116) against `instances_<split>.json`."""
    with open(pred_json) as f:
        pred = json.load(f)
    with open(gt_json) as f:
        gt = json.load(f)
    ids = set(image_ids) if image_ids is not None else {im["id"] for im in pred.get("images", [])} or {a["image_id"] for a in pred["annotations"]}
    gt_by, dt_by = {}, {}
    for a in gt.get("annotations", []):
        if a["image_id"] in ids:
            b = [float(v) for v in a["bbox"]]
            gt_by.setdefault(a["image_id"], []).append({"bbox": b, "area": float(a.get("area", b[2] * b[3])), "iscrowd": a.get("iscrowd", 0),
                                                        "ignore": a.get("ignore", 0)})
    for a in pred.get("annotations", []):
        if a["image_id"] in ids:
            b = reference_box(a["bbox"])
            dt_by.setdefault(a["image_id"], []).append({"bbox": b, "score": float(a["score"]), "area": float(b[2] * b[3])})
    return summarize(gt_by, dt_by)
