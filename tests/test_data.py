"""FSC-147 readers / collate / counting evaluator against golden vectors produced by the REAL
reference dataset classes on the tiny dataset committed under tests/golden/fsc147_tiny (oracle/gen_golden_data.py)."""
import argparse
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
DS = os.path.join(HERE, "golden", "fsc147_tiny")


@pytest.fixture(scope="module")
def args():
    return argparse.Namespace(data_path=DS, scale_factor=32)


def _check(sample, z, prefix):
    keys = [k.split("/", 1)[1] for k in z.files if k.startswith(prefix + "/")]
    assert sorted(keys) == sorted(sample.keys()), (sorted(keys), sorted(sample.keys()))
    for k in keys:
        ref = z[f"{prefix}/{k}"]
        got = sample[k].numpy() if torch.is_tensor(sample[k]) else np.asarray(sample[k])
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        if ref.dtype.kind == "f":
            np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6, err_msg=f"{prefix}/{k}")
        else:
            assert np.array_equal(got, ref), f"{prefix}/{k}"


def test_train_reader_matches_reference(golden, args):
    from counting_detr_amd.data import build_dataset
    z = golden("g9_data.npz")
    ds = build_dataset(args)
    assert len(ds) == 2
    for i in range(len(ds)):
        s = ds[i]
        assert s["image"].shape[1] % 32 == 0 and s["image"].shape[2] % 32 == 0
        _check(s, z, f"train{i}")


@pytest.mark.parametrize("split,n", [("val", 2), ("test", 1)])
def test_eval_readers_match_reference(golden, args, split, n):
    from counting_detr_amd.data import build_test_dataset
    z = golden("g9_data.npz")
    ds = build_test_dataset(args, image_set=split)
    assert len(ds) == n
    for i in range(n):
        _check(ds[i], z, f"{split}{i}")


def test_fscd_lvis_readers_match_reference(golden):
    """BASELINE config 4: same network, FSCD-LVIS files (L2/data/fscd_lvis.py); incl. the train-only exemplar clipping."""
    from counting_detr_amd.data import FSCDLVISDataset
    z = golden("g9_data.npz")
    a = argparse.Namespace(data_path=os.path.join(HERE, "golden", "fscd_lvis_tiny"))
    tr = FSCDLVISDataset(a, split="train")
    assert len(tr) == 2
    for i in range(2):
        _check(tr[i], z, f"lvis_train{i}")
    te = FSCDLVISDataset(a, split="test", test=True)
    assert len(te) == 1
    _check(te[0], z, "lvis_test0")


def test_collate_pads_and_masks(args):
    from counting_detr_amd.data import build_dataset, collate
    ds = build_dataset(args)
    s0, s1 = ds[0], ds[1]
    b = collate([s0, s1])
    H = max(s0["image"].shape[1], s1["image"].shape[1]); W = max(s0["image"].shape[2], s1["image"].shape[2])
    assert b["image"].shape == (2, 3, H, W) and b["mask"].shape == (2, H, W) and b["mask"].dtype == torch.bool
    for i, s in enumerate((s0, s1)):
        h, w = s["image"].shape[1:]
        assert torch.equal(b["image"][i, :, :h, :w], s["image"])
        assert not b["mask"][i, :h, :w].any() and b["mask"][i, h:].all() and b["mask"][i, :, w:].all()
        assert float(b["image"][i, :, h:].abs().sum()) == 0.0
        assert torch.equal(b["targets"][i]["boxes"], torch.as_tensor(s["boxes"])) and b["targets"][i]["labels"].dtype == torch.int64
    assert b["ex_rects"].shape == (2, 3, 4)


def test_prefetcher_cpu_passthrough(args):
    from torch.utils.data import DataLoader
    from counting_detr_amd.data import Prefetcher, build_dataset, collate
    dl = DataLoader(build_dataset(args), batch_size=1, shuffle=False, collate_fn=collate)
    a = [b["image"].sum().item() for b in dl]
    p = Prefetcher(dl, "cpu")
    assert len(p) == len(dl)
    assert [b["image"].sum().item() for b in p] == a


def test_counting_metrics_from_json(tmp_path, args):
    """MAE / RMSE / NAE / SRE as A2/eval_all.py:252-270 on a hand-made predictions json."""
    from infer import counting_metrics_from_json
    gt = os.path.join(DS, "instances_val.json")
    ids = [im["id"] for im in json.load(open(gt))["images"]]
    ngt = {i: sum(1 for a in json.load(open(gt))["annotations"] if a["image_id"] == i) for i in ids}
    pred = {"categories": [{"name": "fg", "id": 1}], "images": [{"id": i, "height": 1, "width": 1, "file_name": "None"} for i in ids],
            "annotations": []}
    want = {ids[0]: ngt[ids[0]] + 2, ids[1]: max(ngt[ids[1]] - 3, 0)}
    k = 1
    for i, c in want.items():
        for _ in range(c):
            pred["annotations"].append({"id": k, "image_id": i, "bbox": [1, 1, 1, 1], "score": 0.9, "category_id": 1, "area": 1, "point": [0, 0]}); k += 1
        pred["annotations"].append({"id": k, "image_id": i, "bbox": [1, 1, 1, 1], "score": 0.2, "category_id": 1, "area": 1, "point": [0, 0]}); k += 1
    pj = tmp_path / "p.json"
    pj.write_text(json.dumps(pred))
    m = counting_metrics_from_json(str(pj), gt)
    errs = np.array([abs(ngt[i] - want[i]) for i in ids], dtype=np.float64)
    g = np.array([ngt[i] for i in ids], dtype=np.float64)
    np.testing.assert_allclose(m["MAE"], errs.mean())
    np.testing.assert_allclose(m["RMSE"], np.sqrt((errs ** 2).mean()))
    np.testing.assert_allclose(m["NAE"], (errs / g).mean())
    np.testing.assert_allclose(m["SRE"], np.sqrt((errs ** 2 / g).mean()))
