"""infer.py -- the reference's inference driver (A2/infer.py:27-122) + the counting part of its evaluator
(A2/eval_all.py:141-279) on the MI355X path.

For every image of the val / test split: forward, losses (logged), the counting rule `sigmoid(logit[..., 0]) >= 0.5`,
predictions written in the reference's wire format (COCO-style json: bbox = [cx, cy, w, h] ints in original pixels, `point` =
the query's reference point, one `images` entry per image) to <output_dir>/predictions_<split>.json, then MAE / RMSE / NAE /
SRE of the predicted counts against the ground-truth instance counts, and -- when the split's `instances_<split>.json` is there -- the box AP /
AP50 / AP75 / APs / APm / APl of A2/eval_all.py:285-331 from a dependency-free restatement of pycocotools' COCOeval
(counting_detr_amd/coco_ap.py; parity unpinned: there is no pycocotools in this image to check it against).

  python infer.py -dp /data/FSC147 --split val --resume out/detr_retrain.pth -o out
"""
import json
import os

import torch

import counting_detr_amd
from counting_detr_amd import data
from counting_detr_amd.args import get_args_parser
from counting_detr_amd.engine import count_from_logits, counting_metrics
from counting_detr_amd.misc import NestedTensor


@torch.no_grad()
def infer(model, criterion, data_loader, device, output_dir, split="test", threshold=0.5, graphs=True):
    """-> (metrics dict, predictions dict); writes predictions_<split>.json like A2/infer.py:28-121.  The forward + counting rule
    runs through engine.InferenceEngine (pre-split weight images, one captured HIP graph per image shape; `graphs=False`: eager)."""
    output_path = os.path.join(output_dir, "predictions_" + split + ".json")
    if os.path.isfile(output_path):
        os.remove(output_path)
    model.eval()
    criterion.eval()
    from counting_detr_amd.engine import InferenceEngine
    engine = InferenceEngine(model, threshold, graphs=graphs and torch.device(device).type == "cuda", device=device)
    predictions = {"categories": [{"name": "fg", "id": 1}], "images": [], "annotations": []}
    anno_id = 1
    pred_counts, gt_counts, loss_sum, n_img = [], [], {}, 0
    def lookahead(loader):          # (batch on the device, the next batch's image tensor or None): the engine runs the next image's frozen
        prev = None                 # stage (stem + layer1) beside this image's encoder / decoder
        for cur in loader:
            cur = dict(cur)
            cur["image"], cur["mask"] = cur["image"].to(device), cur["mask"].to(device)
            if prev is not None:
                yield prev, cur["image"]
            prev = cur
        if prev is not None:
            yield prev, None

    for ret, next_image in lookahead(data_loader):
        image, mask = ret["image"], ret["mask"]
        rects = ret["ex_rects"].to(device)
        targets = [{k: v.to(device) for k, v in t.items()} for t in ret["targets"]]
        _, keep, outputs, ref_points, prob = engine(NestedTensor(image, mask), rects, next_samples=next_image)      # forward + :75-81
        loss_dict = criterion(outputs, targets)
        for k, v in loss_dict.items():
            loss_sum[k] = loss_sum.get(k, 0.0) + float(v) * len(targets)
        for b in range(image.shape[0]):
            ori_h, ori_w = [int(x) for x in ret["orig_size"][b]]
            image_id = int(ret["image_id"][b]) if "image_id" in ret else n_img
            kb = keep[b]
            scores = prob[b][kb].cpu().numpy()
            boxes = outputs["pred_boxes"][b][kb].cpu().numpy().copy()
            pts = ref_points[b][kb].cpu().numpy().copy()
            pts[..., 0] *= ori_w; pts[..., 1] *= ori_h
            boxes[..., 0] *= ori_w; boxes[..., 1] *= ori_h; boxes[..., 2] *= ori_w; boxes[..., 3] *= ori_h
            for sc, bx, pt in zip(scores, boxes, pts):
                x_cen, y_cen, w, h = bx
                predictions["annotations"].append({"id": anno_id, "image_id": image_id, "area": int(w * h),
                                                   "bbox": [int(x_cen), int(y_cen), int(w), int(h)], "category_id": 1,
                                                   "score": float(sc), "point": [int(pt[0]), int(pt[1])]})
                anno_id += 1
            predictions["images"].append({"id": image_id, "height": ori_h, "width": ori_w, "file_name": "None"})
            pred_counts.append(int(kb.sum()))
            gt_counts.append(int(targets[b]["boxes"].shape[0]))
            n_img += 1
    with open(output_path, "w") as handle:
        json.dump(predictions, handle)
    metrics = {k: v / max(n_img, 1) for k, v in loss_sum.items()}
    if n_img:       # images without objects contribute to MAE / RMSE only (the reference divides by the count, A2/eval_all.py:264-265)
        metrics.update(counting_metrics(pred_counts, gt_counts))
    metrics["images"] = n_img
    return metrics, predictions


def counting_metrics_from_json(pred_json, gt_json, threshold=0.5):
    """MAE / RMSE / NAE / SRE from a predictions json and the split's `instances_<split>.json` (A2/eval_all.py:141-270:
    predicted count = #annotations with score >= threshold per image, ground truth = #instances)."""
    with open(pred_json) as f:
        pred = json.load(f)
    gt = data.CocoIndex(gt_json)
    cnt = {im["id"]: 0 for im in pred["images"]}
    for a in pred["annotations"]:
        if a["score"] >= threshold:
            cnt[a["image_id"]] = cnt.get(a["image_id"], 0) + 1
    ids = sorted(cnt)
    return counting_metrics([cnt[i] for i in ids], [len(gt.getAnnIds([i])) for i in ids])


def main(args):
    device = torch.device(args.device)
    model, criterion, _ = counting_detr_amd.build_model(args)
    model.to(device)
    if args.resume:
        ckpt = torch.load(args.resume, map_location="cpu", weights_only=False)
        model.load_state_dict(ckpt["model"], strict=True)
    from torch.utils.data import DataLoader
    ds = data.build_test_dataset(args, image_set=args.split)
    dl = DataLoader(ds, batch_size=1, shuffle=False, collate_fn=data.collate, num_workers=args.num_workers)
    os.makedirs(args.output_dir, exist_ok=True)
    metrics, _ = infer(model, criterion, dl, device, args.output_dir, split=args.split)
    gt_json = os.path.join(args.data_path, "instances_" + args.split + ".json")
    if os.path.isfile(gt_json):
        from counting_detr_amd.coco_ap import ap_from_json
        metrics.update(ap_from_json(os.path.join(args.output_dir, "predictions_" + args.split + ".json"), gt_json))
    print(json.dumps(metrics))
    with open(os.path.join(args.output_dir, "results_" + args.split + ".txt"), "w") as f:
        f.write(json.dumps(metrics) + "\n")


if __name__ == "__main__":
    main(get_args_parser().parse_args())
