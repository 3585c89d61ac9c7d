"""Golden vectors of the reference's FSC-147 readers (TEST INFRA).

    python -m oracle.gen_golden_data       # needs /root/reference; writes tests/golden/fsc147_tiny/ + g9_data.npz

A tiny FSC-147-shaped dataset (4 noise images with odd sizes, the five json files the readers open) is written under
tests/golden/fsc147_tiny/ and the REAL reference classes (A2/data/fsc147.py: FSC147Dataset, FSC147_Dataset_Val,
FSC147_Dataset_Test) are run on it.  Two third-party imports of that file are absent from this image and are replaced by
stand-ins written here (so parity of exactly these two pieces is by definition, not by the reference's dependency):
  * pycocotools.coco.COCO        -> an index over the json with the five calls the readers make;
  * torchvision.transforms       -> Compose / ToTensor (uint8 HWC -> float32 CHW / 255) / Normalize ((x - mean) / std).
Everything else (file layout, resize rule and PIL filter, box / exemplar / point normalisation, field names, dtypes) is
the reference's own code.  The fixture holds the dataset files + expected outputs (data), no reference source.
"""
import json
import os
import sys
import types

import numpy as np
import torch
from PIL import Image

REF = "/root/reference/src/CountDETR_147_2nd_stage"
ROOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
DS = os.path.join(ROOT, "fsc147_tiny")


def write_dataset():
    rng = np.random.default_rng(7)
    os.makedirs(os.path.join(DS, "images_384_VarV2"), exist_ok=True)
    os.makedirs(os.path.join(DS, "annotations"), exist_ok=True)
    sizes = {"1.jpg": (101, 70), "2.jpg": (97, 131), "3.jpg": (64, 64), "4.jpg": (150, 99)}      # (w, h)
    anno, coco = {}, {n: {"images": [], "annotations": [], "categories": [{"id": 1, "name": "fg"}]} for n in ("train", "val", "test")}
    split = {"train": ["1.jpg", "2.jpg"], "val": ["3.jpg", "1.jpg"], "test": ["4.jpg"]}
    aid = 1
    for i, (name, (w, h)) in enumerate(sizes.items(), start=1):
        # smooth-ish content so the resize filters differ measurably
        base = rng.integers(0, 255, (h // 4 + 2, w // 4 + 2, 3), dtype=np.uint8)
        img = Image.fromarray(base).resize((w, h), Image.BICUBIC)
        img.save(os.path.join(DS, "images_384_VarV2", name.replace(".jpg", ".png")))
        ex = []
        for _ in range(3):
            x1, y1 = rng.uniform(0, w * 0.6), rng.uniform(0, h * 0.6)
            x2, y2 = x1 + rng.uniform(4, w * 0.3), y1 + rng.uniform(4, h * 0.3)
            ex.append([[float(x1), float(y1)], [float(x1), float(y2)], [float(x2), float(y2)], [float(x2), float(y1)]])
        pts = [[float(rng.uniform(0, w)), float(rng.uniform(0, h))] for _ in range(5 + i)]
        anno[name.replace(".jpg", ".png")] = {"box_examples_coordinates": ex, "points": pts, "H": h, "W": w}
    for sp, names in split.items():
        for name in names:
            png = name.replace(".jpg", ".png")
            iid = int(name.split(".")[0]) + (100 if sp == "train" else 0)
            w, h = sizes[name]
            coco[sp]["images"].append({"id": iid, "file_name": png, "width": w, "height": h})
            for p in anno[png]["points"]:
                bw, bh = float(rng.uniform(3, 12)), float(rng.uniform(3, 12))
                bbox = [p[0], p[1], bw, bh] if sp == "train" else [p[0] - bw / 2, p[1] - bh / 2, bw, bh]
                coco[sp]["annotations"].append({"id": aid, "image_id": iid, "bbox": bbox, "category_id": 1, "area": bw * bh, "iscrowd": 0})
                aid += 1
    with open(os.path.join(DS, "annotation_FSC147_384.json"), "w") as f:
        json.dump(anno, f)
    with open(os.path.join(DS, "Train_Test_Val_FSC_147.json"), "w") as f:
        json.dump({k: [n.replace(".jpg", ".png") for n in v] for k, v in split.items()}, f)
    with open(os.path.join(DS, "annotations", "pseudo_bbox_train.json"), "w") as f:
        json.dump(coco["train"], f)
    for sp in ("val", "test"):
        with open(os.path.join(DS, f"instances_{sp}.json"), "w") as f:
            json.dump(coco[sp], f)


def install_stubs():
    class COCO:
        def __init__(self, path):
            d = json.load(open(path))
            self.imgs = {im["id"]: im for im in d["images"]}
            self.anns = {a["id"]: a for a in d["annotations"]}
            self.by = {}
            for a in d["annotations"]:
                self.by.setdefault(a["image_id"], []).append(a["id"])

        def getImgIds(self):
            return list(self.imgs.keys())

        def loadImgs(self, ids):
            return [self.imgs[i] for i in ids]

        def getAnnIds(self, ids):
            return [a for i in ids for a in self.by.get(i, [])]

        def loadAnns(self, ids=()):
            return [self.anns[i] for i in ids]

    pc, pcc = types.ModuleType("pycocotools"), types.ModuleType("pycocotools.coco")
    pcc.COCO = COCO
    sys.modules["pycocotools"], sys.modules["pycocotools.coco"] = pc, pcc

    class ToTensor:
        def __call__(self, img):
            a = np.asarray(img.convert("RGB"), dtype=np.uint8)
            return torch.from_numpy(a).permute(2, 0, 1).float().div(255.0)

    class Normalize:
        def __init__(self, mean, std):
            self.m, self.s = torch.tensor(mean).view(3, 1, 1), torch.tensor(std).view(3, 1, 1)

        def __call__(self, t):
            return (t - self.m) / self.s

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    tv, tvt = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")
    tvt.ToTensor, tvt.Normalize, tvt.Compose = ToTensor, Normalize, Compose
    tv.transforms = tvt
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tvt


def main():
    write_dataset()
    install_stubs()
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_fsc147", os.path.join(REF, "data", "fsc147.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    import argparse
    args = argparse.Namespace(data_path=DS, scale_factor=32)
    d = {}

    def dump(prefix, sample):
        for k, v in sample.items():
            if isinstance(v, torch.Tensor):
                v = v.numpy()
            d[f"{prefix}/{k}"] = np.asarray(v)

    tr = ref.FSC147Dataset(args, split="train")
    for i in range(len(tr)):
        dump(f"train{i}", tr[i])
    va = ref.FSC147_Dataset_Val(args, split="val")
    for i in range(len(va)):
        dump(f"val{i}", va[i])
    te = ref.FSC147_Dataset_Test(args, split="test")
    for i in range(len(te)):
        dump(f"test{i}", te[i])
    # ---- FSCD-LVIS readers (L2/data/fscd_lvis.py) on the same images
    LV = os.path.join(ROOT, "fscd_lvis_tiny")
    os.makedirs(os.path.join(LV, "images", "all_images"), exist_ok=True)
    os.makedirs(os.path.join(LV, "annotations_old"), exist_ok=True)
    import shutil
    rng = np.random.default_rng(11)
    for sp, names in (("train", ["1.png", "2.png"]), ("test", ["4.png"])):
        coco = {"images": [], "annotations": [], "categories": [{"id": 1, "name": "fg"}]}
        cnt = {"annotations": []}
        aid = 1
        for i, n in enumerate(names, start=1):
            shutil.copy(os.path.join(DS, "images_384_VarV2", n), os.path.join(LV, "images", "all_images", n))
            w, h = Image.open(os.path.join(DS, "images_384_VarV2", n)).size
            coco["images"].append({"id": 10 * i, "file_name": n, "width": w, "height": h})
            for _ in range(4 + i):
                coco["annotations"].append({"id": aid, "image_id": 10 * i, "category_id": 1, "iscrowd": 0,
                                            "bbox": [float(rng.uniform(0, w)), float(rng.uniform(0, h)), float(rng.uniform(3, 15)), float(rng.uniform(3, 15))]})
                aid += 1
            # exemplar boxes, one of them sticking out of the image (clipped on the training split only)
            cnt["annotations"].append({"boxes": [[float(rng.uniform(0, w * 0.5)), float(rng.uniform(0, h * 0.5)), float(rng.uniform(5, 30)), float(rng.uniform(5, 30))] for _ in range(3)]
                                       + [[w - 5.0, h - 4.0, 20.0, 20.0]]})
            cnt["annotations"][-1]["boxes"][1] = [w - 6.0, h - 7.0, 25.0, 25.0]
        fn = "pseudo_lvis_train_cxcywh.json" if sp == "train" else "single_instances_test.json"
        json.dump(coco, open(os.path.join(LV, "annotations_old", fn), "w"))
        json.dump(cnt, open(os.path.join(LV, "annotations_old", f"count_{sp}.json"), "w"))
    spec = importlib.util.spec_from_file_location("ref_lvis", "/root/reference/src/CountDETR_lvis_2nd_stage/data/fscd_lvis.py")
    lv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lv)
    largs = argparse.Namespace(data_path=LV)
    tr = lv.FSCD_LVISDataset(largs, split="train")
    for i in range(len(tr)):
        dump(f"lvis_train{i}", tr[i])
    te = lv.FSCD_LVIS_Dataset_Test(largs, split="test")
    for i in range(len(te)):
        dump(f"lvis_test{i}", te[i])
    np.savez_compressed(os.path.join(ROOT, "g9_data.npz"), **d)
    print("wrote", len(d), "arrays;", sorted(set(k.split("/")[0] for k in d)))


if __name__ == "__main__":
    main()
