"""FSC-147 readers + batched collate for the 2nd-stage step.

Sample semantics follow the reference's dataset classes field by field:
  train  A2/data/fsc147.py:12-102   pseudo-label COCO json `annotations/pseudo_bbox_<split>.json` (bbox = [cx, cy, w, h] in pixels),
                                    exemplar rectangles from `annotation_FSC147_384.json`, image resized to floor(w/32)*32 x
                                    floor(h/32)*32 with PIL's default filter, ToTensor + ImageNet normalisation, boxes / rects
                                    divided by (w, h, w, h);
  val    A2/data/fsc147.py:105-211  `instances_val.json` (bbox = [x1, y1, w, h]) -> centre boxes, points, xyxy boxes; resize to a
  test   A2/data/fsc147.py:214-351  multiple of `scale_factor` with BILINEAR.
Differences (MI355X-first): no pycocotools (the COCO json is indexed directly); any number of images per step
(`collate` pads to the batch maximum and builds the padding mask the model consumes, the reference is batch-1 only);
`Prefetcher` stages the next batch through pinned memory on a side stream while the current step runs (the pattern the
reference sketches in A1/datasets/data_prefetcher.py:23-79 and never uses).
"""
import json
import os

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def to_normalized_tensor(img):
    """transforms.ToTensor() + Normalize(ImageNet) of a PIL image -> float32 [3,H,W] (A2/data/fsc147.py:22-24)."""
    a = np.array(img.convert("RGB"), dtype=np.uint8)
    t = torch.from_numpy(a).permute(2, 0, 1).to(torch.float32).div(255.0)
    return (t - torch.from_numpy(MEAN).view(3, 1, 1)) / torch.from_numpy(STD).view(3, 1, 1)


class CocoIndex:
    """The four pycocotools.COCO calls the reference uses (getImgIds / loadImgs / getAnnIds / loadAnns / .imgs)."""

    def __init__(self, path):
        with open(path, "r") as f:
            d = json.load(f)
        self.imgs = {im["id"]: im for im in d.get("images", [])}
        self.anns = {}
        self._by_img = {}
        for a in d.get("annotations", []):
            self.anns[a["id"]] = a
            self._by_img.setdefault(a["image_id"], []).append(a["id"])

    def getImgIds(self):
        return list(self.imgs.keys())

    def loadImgs(self, ids):
        return [self.imgs[i] for i in ids]

    def getAnnIds(self, img_ids):
        out = []
        for i in img_ids:
            out += self._by_img.get(i, [])
        return out

    def loadAnns(self, ids):
        return [self.anns[i] for i in ids]


def _load_json(path):
    with open(path, "r") as f:
        return json.load(f)


def _exemplar_rects(anno):
    """box_examples_coordinates: four corner points per exemplar -> [x1, y1, x2, y2] (A2/data/fsc147.py:52-61)."""
    return np.array([[b[0][0], b[0][1], b[2][0], b[2][1]] for b in anno["box_examples_coordinates"]], dtype=np.float32)


class FSC147Dataset(Dataset):
    """Training split (A2/data/fsc147.py:12-102)."""

    def __init__(self, args, split="train"):
        data_path = args.data_path
        self.coco = CocoIndex(os.path.join(data_path, "annotations", "pseudo_bbox_" + split + ".json"))
        self.images = self.coco.getImgIds()
        self.img_path = os.path.join(data_path, "images_384_VarV2")
        self.annotations = _load_json(os.path.join(data_path, "annotation_FSC147_384.json"))

    def __len__(self):
        return len(self.images)

    def __getitem__(self, index):
        img_info = self.coco.loadImgs([self.images[index]])[0]
        img_file = img_info["file_name"]
        img = Image.open(os.path.join(self.img_path, img_file))
        wh = img.size
        anns = self.coco.loadAnns(self.coco.getAnnIds([self.images[index]]))
        bboxes = np.array([a["bbox"] for a in anns], dtype=np.float32).reshape(-1, 4)
        ex_rects = _exemplar_rects(self.annotations[img_file])
        img_w, img_h = img.size
        img = img.resize((32 * int(img_w / 32), 32 * int(img_h / 32)))                  # :75-77 (PIL default filter)
        res = np.array([img_w, img_h, img_w, img_h], dtype=np.float32)
        bboxes = bboxes / res[None, :]
        xyxy = np.zeros_like(bboxes)
        xyxy[:, 0], xyxy[:, 1] = bboxes[:, 0] - bboxes[:, 2] / 2, bboxes[:, 1] - bboxes[:, 3] / 2
        xyxy[:, 2], xyxy[:, 3] = bboxes[:, 0] + bboxes[:, 2] / 2, bboxes[:, 1] + bboxes[:, 3] / 2
        return {"image": to_normalized_tensor(img), "boxes": bboxes, "ex_rects": ex_rects / res[None, :], "origin_wh": wh,
                "labels": torch.zeros([bboxes.shape[0]], dtype=torch.int64), "orig_size": np.array([img_h, img_w]),
                "xyxy_boxes": xyxy}


class FSC147EvalDataset(Dataset):
    """Validation / test split (A2/data/fsc147.py:105-211, :214-351): `instances_<split>.json` ground truth."""

    def __init__(self, args, split="val"):
        data_path = args.data_path
        self.im_dir = os.path.join(data_path, "images_384_VarV2")
        self.scale_factor = args.scale_factor
        self.annotations = _load_json(os.path.join(data_path, "annotation_FSC147_384.json"))
        self.data_split = _load_json(os.path.join(data_path, "Train_Test_Val_FSC_147.json"))[split]
        self.label = CocoIndex(os.path.join(data_path, f"instances_{split}.json"))
        self.name2id = {v["file_name"]: v["id"] for v in self.label.imgs.values()}

    def __len__(self):
        return len(self.data_split)

    def __getitem__(self, idx):
        name = self.data_split[idx]
        im_id = self.name2id[name]
        annos = self.label.loadAnns(self.label.getAnnIds([im_id]))
        centers = np.array([[a["bbox"][0] + a["bbox"][2] / 2, a["bbox"][1] + a["bbox"][3] / 2] for a in annos], dtype=np.float32).reshape(-1, 2)
        whs = np.array([[a["bbox"][2], a["bbox"][3]] for a in annos], dtype=np.float32).reshape(-1, 2)
        xyxy = np.array([[a["bbox"][0], a["bbox"][1], a["bbox"][0] + a["bbox"][2], a["bbox"][1] + a["bbox"][3]] for a in annos],
                        dtype=np.float32).reshape(-1, 4)
        ex = _exemplar_rects(self.annotations[name])
        image = Image.open("{}/{}".format(self.im_dir, name))
        img_w, img_h = image.size
        res4 = np.array([img_w, img_h, img_w, img_h], dtype=np.float32)
        sf = self.scale_factor
        image = image.resize((sf * int(img_w / sf), sf * int(img_h / sf)), Image.BILINEAR)
        return {"image_id": im_id, "image": to_normalized_tensor(image), "points": centers / res4[None, :2],
                "boxes": np.concatenate((centers, whs), axis=1) / res4[None, :], "orig_size": np.array([img_h, img_w]),
                "exemplar_boxes": ex / res4[None, :], "labels": np.zeros(centers.shape[0], dtype=np.int64),
                "xyxy_boxes": xyxy / res4[None, :]}


class FSCDLVISDataset(Dataset):
    """FSCD-LVIS train / test readers (BASELINE config 4; L2/data/fscd_lvis.py:12-100, :103-190): same network, different
    files -- `annotations_old/pseudo_lvis_<split>_cxcywh.json` (train) or `single_instances_<split>.json` (test), exemplars
    from `count_<split>.json` (first three [x, y, w, h] boxes, clipped to the image on the training split only), RGB convert."""

    def __init__(self, args, split="train", test=False):
        data_path = args.data_path
        name = ("single_instances_" + split + ".json") if test else ("pseudo_lvis_" + split + "_cxcywh.json")
        self.coco = CocoIndex(os.path.join(data_path, "annotations_old", name))
        self.image_ids = self.coco.getImgIds()
        self.img_path = os.path.join(data_path, "images", "all_images")
        self.count_anno = _load_json(os.path.join(data_path, "annotations_old", "count_" + split + ".json"))
        self.clip = not test

    def __len__(self):
        return len(self.image_ids)

    def __getitem__(self, idx):
        img_id = self.image_ids[idx]
        img_file = self.coco.loadImgs([img_id])[0]["file_name"]
        img = Image.open(os.path.join(self.img_path, img_file)).convert("RGB")
        wh = img.size
        anns = self.coco.loadAnns(self.coco.getAnnIds([img_id]))
        bboxes = np.array([a["bbox"] for a in anns], dtype=np.float32).reshape(-1, 4)
        ex = np.array([[x, y, x + w, y + h] for x, y, w, h in self.count_anno["annotations"][idx]["boxes"][:3]], dtype=np.float32)
        if self.clip:                                                                    # L2/data/fscd_lvis.py:60-63
            ex[:, 0] = np.clip(ex[:, 0], 0, wh[0] - 1); ex[:, 1] = np.clip(ex[:, 1], 0, wh[1] - 1)
            ex[:, 2] = np.clip(ex[:, 2], 0, wh[0] - 1); ex[:, 3] = np.clip(ex[:, 3], 0, wh[1] - 1)
        img_w, img_h = wh
        img = img.resize((32 * int(img_w / 32), 32 * int(img_h / 32)))
        res = np.array([img_w, img_h, img_w, img_h], dtype=np.float32)
        bboxes = bboxes / res[None, :]
        return {"image": to_normalized_tensor(img), "boxes": bboxes, "ex_rects": ex / res[None, :], "origin_wh": wh,
                "labels": torch.zeros([bboxes.shape[0]], dtype=torch.int64), "orig_size": np.array([img_h, img_w])}


def build_dataset(args):
    if getattr(args, "dataset", "fsc147") == "fscd_lvis":
        return FSCDLVISDataset(args, split="train")
    return FSC147Dataset(args)


def build_test_dataset(args, image_set="val"):
    return FSC147EvalDataset(args, split="val" if image_set == "val" else "test")


def collate(samples):
    """List of dataset samples -> the step's batch dict: images padded to the batch maximum with the padding mask
    (NestedTensor convention: True = padding), exemplar rectangles [B,3,4], per-image target dicts."""
    B = len(samples)
    Hm = max(s["image"].shape[1] for s in samples)
    Wm = max(s["image"].shape[2] for s in samples)
    image = torch.zeros((B, 3, Hm, Wm), dtype=torch.float32)
    mask = torch.ones((B, Hm, Wm), dtype=torch.bool)
    for b, s in enumerate(samples):
        _, h, w = s["image"].shape
        image[b, :, :h, :w] = s["image"]
        mask[b, :h, :w] = False
    rk = "ex_rects" if "ex_rects" in samples[0] else "exemplar_boxes"
    rl = [torch.as_tensor(s[rk], dtype=torch.float32).reshape(-1, 4)[:3] for s in samples]
    # FSCD-LVIS has "at most 3" exemplars (L2/data/fscd_lvis.py:53): absent rows are marked with -1 (backbone per_image mode skips them)
    rects = torch.stack([torch.cat([r, torch.full((3 - r.shape[0], 4), -1.0)]) if r.shape[0] < 3 else r for r in rl])
    targets = [{"boxes": torch.as_tensor(s["boxes"], dtype=torch.float32).reshape(-1, 4),
                "labels": torch.as_tensor(s["labels"], dtype=torch.int64).reshape(-1)} for s in samples]
    out = {"image": image, "mask": mask, "ex_rects": rects, "targets": targets,
           "orig_size": torch.as_tensor(np.stack([np.asarray(s["orig_size"]) for s in samples]))}
    if "image_id" in samples[0]:
        out["image_id"] = torch.as_tensor([int(s["image_id"]) for s in samples])
    return out


class Prefetcher:
    """Iterates a DataLoader of collated batches one step ahead: the next batch is copied host -> device through pinned
    memory on a side stream while the current step computes; `next()` hands over device tensors after an event wait."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def _to_device(self, batch):
        def mv(t):
            if not torch.is_tensor(t):
                return t
            if self.stream is not None:
                t = t.pin_memory() if not t.is_pinned() else t
            return t.to(self.device, non_blocking=True)
        out = {k: mv(v) for k, v in batch.items() if k != "targets"}
        out["targets"] = [{k: mv(v) for k, v in t.items()} for t in batch["targets"]]
        return out

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        it = iter(self.loader)

        def stage():
            try:
                b = next(it)
            except StopIteration:
                return None
            if self.stream is None:
                return self._to_device(b)
            with torch.cuda.stream(self.stream):
                return self._to_device(b)

        nxt = stage()
        while nxt is not None:
            if self.stream is not None:
                torch.cuda.current_stream(self.device).wait_stream(self.stream)
                for v in list(nxt.values()) + [x for t in nxt["targets"] for x in t.values()]:
                    if torch.is_tensor(v):
                        v.record_stream(torch.cuda.current_stream(self.device))
            cur, nxt = nxt, stage()
            yield cur
