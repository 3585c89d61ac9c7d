"""The graph-cached train step (engine.Trainer.step): one captured HIP graph per padded image size and
target-capacity class serves batches of ANY target counts -- every count-dependent quantity of the matcher / criterion is a
device table (ops.MatchPlan.capacity + ops.PackedTargets).  Reference loop: A2/engine.py:14-67 fed by A2/data/fsc147.py:69-102
(variable image widths, 7...3731 targets per image).

CPU part: the capacity plan's tables.  GPU part (-m gpu): capacity plan == exact plan through the device matcher + criterion
(indices bit-exact), cached-graph steps == stream-ordered steps on fresh batches of different sizes / counts, LRU eviction,
and main.py on a generated FSC-147-format dataset with several image sizes."""
import json
import os

import numpy as np
import pytest
import torch

from counting_detr_amd import ops

HAS_GPU = torch.cuda.is_available()
DEV = "cuda:0"


def test_capacity_plan_tables_cpu():
    p = ops.MatchPlan.capacity(3, 300, 128, "cpu")
    assert (p.Mmax, p.nc_max, p.cost_numel) == (128, 300, 3 * 300 * 128)
    assert p.cost_off.tolist() == [0, 300 * 128, 2 * 300 * 128]
    p.set_counts([5, 0, 128])
    assert p.tgt_off.tolist() == [0, 5, 5, 133] and p.sizes_f.tolist() == [5.0, 0.0, 128.0] and p.M == [5, 0, 128]
    with pytest.raises(AssertionError):
        p.set_counts([129, 0, 0])
    q = ops.MatchPlan.capacity(2, 300, 512, "cpu")              # more targets than queries: the matrix is [Q][T], Q rows get matched
    assert (q.Mmax, q.nc_max) == (300, 512)
    e = ops.MatchPlan([5, 0, 128], 300, "cpu")                  # the exact plan of the same counts packs the cost matrices tightly
    assert e.tgt_off.tolist() == p.tgt_off.tolist() and e.Mmax == 128 and e.cost_off.tolist() == [0, 1500, 1500]


def test_packed_targets_load_cpu():
    pk = ops.PackedTargets.with_capacity(2, 300, 128, "cpu")
    tg = [{"boxes": torch.rand(3, 4), "labels": torch.zeros(3, dtype=torch.int64)},
          {"boxes": torch.rand(0, 4), "labels": torch.zeros(0, dtype=torch.int64)}]
    assert pk.load(tg) == 3
    assert torch.equal(pk.boxes[:3], tg[0]["boxes"]) and pk.plan.tgt_off.tolist() == [0, 3, 3]
    tg2 = [{"boxes": torch.rand(1, 4), "labels": torch.ones(1, dtype=torch.int64)},
           {"boxes": torch.rand(2, 4), "labels": torch.zeros(2, dtype=torch.int64)}]
    ptr = pk.boxes.data_ptr()
    pk.load(tg2)
    assert pk.boxes.data_ptr() == ptr and pk.plan.tgt_off.tolist() == [0, 1, 3] and int(pk.labels[0]) == 1
    assert torch.equal(pk.boxes[1:3], tg2[1]["boxes"])


def test_trainer_capacity_classes_cpu():
    from counting_detr_amd.engine import Trainer

    class _T:
        _queries = lambda self: 300          # noqa: E731
    t = _T()
    f = lambda n: Trainer.target_capacity(t, n)      # noqa: E731
    assert [f(n) for n in (0, 7, 128, 129, 300, 301, 512, 513, 3731)] == [128, 128, 128, 300, 300, 512, 512, 1024, 3800]
    with pytest.raises(ValueError):
        f(3801)


# ---------------------------------------------------------------------------------------------------------------- GPU
gpu = pytest.mark.gpu


def _build(Q=100):
    import counting_detr_amd
    from counting_detr_amd.args import default_args
    from counting_detr_amd.init import seeded_init_
    args = default_args(device=DEV, num_query_position=Q)
    model, crit, _ = counting_detr_amd.build_model(args)
    seeded_init_(model)
    model.to(DEV).train()
    crit.train()
    return model, crit, args


def _batch(B, H, W, Ts, seed):
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(B, 3, H, W, generator=g).to(DEV)
    rects = torch.tensor([[.10, .10, .20, .20], [.40, .40, .50, .55], [.70, .20, .80, .30]])[None].repeat(B, 1, 1).to(DEV)
    tg = []
    for b in range(B):
        T = Ts[b]
        box = torch.cat([torch.rand(T, 2, generator=g) * 0.8 + 0.1, torch.rand(T, 2, generator=g) * 0.10 + 0.02], 1)
        tg.append({"boxes": box.to(DEV), "labels": torch.zeros(T, dtype=torch.int64, device=DEV)})
    return images, rects, tg


@gpu
@pytest.mark.parametrize("Ts,cap", [((7, 13), 128), ((0, 31), 128), ((100, 3), 100), ((130, 40), 512), ((250, 0), 512),
                                    ((600, 40), 1024), ((1024, 1024), 1024), ((2000, 3), 2048), ((1025, 0), 2048), ((3731, 100), 3800), ((2049, 3800), 3800)])
def test_capacity_plan_equals_exact_plan_on_device(Ts, cap):
    """Device matcher + fused criterion on PackedTargets with a capacity plan == on the reference's list of dicts with the exact
    plan: Hungarian indices bit-exact (rows beyond min(Q, T) are not part of the contract), every loss equal."""
    from counting_detr_amd.matcher import OriginalHungarianMatcher
    Q, B = 100, 2
    g = torch.Generator().manual_seed(11)
    out = {"pred_logits": torch.randn(B, Q, 2, generator=g).to(DEV), "pred_boxes": (torch.rand(B, Q, 4, generator=g) * 0.5 + 0.2).to(DEV),
           "pred_vars": (torch.rand(B, Q, 2, generator=g) + 0.5).to(DEV)}
    _, _, tg = _batch(B, 32, 32, Ts, seed=3)
    m = OriginalHungarianMatcher(2, 5, 2)
    ii, jj, st, plan = m.match_device(out, tg)
    pk = ops.PackedTargets.with_capacity(B, Q, cap, DEV)
    pk.load(tg)
    ii2, jj2, st2, _ = m.match_device(out, None, pk.plan, tgt_boxes=pk.boxes)
    assert int(st.abs().sum()) == 0 and int(st2.abs().sum()) == 0
    for b in range(B):
        Mb = min(Q, Ts[b])
        assert torch.equal(ii[b, :Mb], ii2[b, :Mb]) and torch.equal(jj[b, :Mb], jj2[b, :Mb])
    _, crit, _ = _build(Q)
    l1 = crit(out, tg)
    l2 = crit(out, pk)
    for k in l1:
        np.testing.assert_allclose(float(l2[k]), float(l1[k]), rtol=1e-6, atol=1e-7, err_msg=k)


@gpu
def test_cached_graph_steps_equal_eager_steps_on_varied_batches():
    """Trainer.step (graph cache) vs Trainer.train_step (stream-ordered) from the SAME weights / moments on six fresh batches of
    three image sizes whose target counts change inside a capacity class and across classes (incl. an image with no target
    and one with more targets than queries): losses to 1e-5, updated parameters to the atomic-order noise of one AdamW step;
    the third size re-uses its captured graph with other counts; captures happen once per (size, class)."""
    from counting_detr_amd.engine import Trainer
    model, crit, args = _build(Q=100)
    tr = Trainer(model, crit, args, device=DEV)
    seq = [((2, 128, 160), (7, 13)), ((2, 96, 128), (30, 5)), ((2, 128, 160), (64, 0)), ((2, 64, 96), (3, 3)),
           ((2, 128, 160), (101, 9)), ((2, 96, 128), (2, 41)), ((2, 128, 160), (1, 100))]
    state = lambda: [t.detach().clone() for t in (tr.flat_p, tr.exp_avg, tr.exp_avg_sq, tr.opt_state)]      # noqa: E731
    for i, ((B, H, W), Ts) in enumerate(seq):
        images, rects, tg = _batch(B, H, W, Ts, seed=100 + i)
        saved = state()
        eo = {k: float(v) for k, v in tr.train_step(images, rects, tg).items()}
        p_eager = tr.flat_p.detach().clone()
        for dst, src in zip((tr.flat_p, tr.exp_avg, tr.exp_avg_sq, tr.opt_state), saved):
            dst.copy_(src)
        go = {k: float(v) for k, v in tr.step(images, rects, tg).items()}
        torch.cuda.synchronize()
        for k in eo:
            np.testing.assert_allclose(go[k], eo[k], rtol=1e-4 if k == "grad_norm" else 1e-5, atol=1e-6, err_msg=f"step {i} {k}")
        diff = (tr.flat_p - p_eager).abs()
        assert float(diff.max()) <= 2.1e-4 and float((diff > 2e-6).float().mean()) < 2e-3, f"step {i}"
    # keys: (128x160, cap 100) x3 [7,13 / 64,0 / 1,100], (96x128, 100) x2, (64x96, 100), (128x160, cap 512) -> 4 captures for 7 steps
    assert tr.cache_stats == {"captures": 4, "steps": 7}
    assert sorted(e["replays"] for e in tr._cache.values()) == [1, 1, 2, 3]


@gpu
def test_cached_graph_steps_with_crowded_images():
    """FSC-147's crowded images (up to 3731 targets per image, A2/data/fsc147.py:80-84): the capacity classes 1024 / 2048 / 3800 of
    the captured step (cost-matrix slots b * Q * Tcap in the [Q][T] layout, the one-workgroup LDS solver) == the stream-ordered
    step with the exact plan; the 2048 graph is replayed with other counts."""
    from counting_detr_amd.engine import Trainer
    model, crit, args = _build(Q=100)
    tr = Trainer(model, crit, args, device=DEV)
    seq = [((2, 64, 96), (1100, 5)), ((2, 64, 96), (37, 2100)), ((2, 64, 96), (3731, 0)), ((2, 64, 96), (1500, 1025)), ((2, 64, 96), (600, 1024))]
    state = lambda: [t.detach().clone() for t in (tr.flat_p, tr.exp_avg, tr.exp_avg_sq, tr.opt_state)]      # noqa: E731
    for i, ((B, H, W), Ts) in enumerate(seq):
        images, rects, tg = _batch(B, H, W, Ts, seed=300 + i)
        saved = state()
        eo = {k: float(v) for k, v in tr.train_step(images, rects, tg).items()}
        p_eager = tr.flat_p.detach().clone()
        for dst, src in zip((tr.flat_p, tr.exp_avg, tr.exp_avg_sq, tr.opt_state), saved):
            dst.copy_(src)
        go = {k: float(v) for k, v in tr.step(images, rects, tg).items()}
        torch.cuda.synchronize()
        for k in eo:
            np.testing.assert_allclose(go[k], eo[k], rtol=1e-4 if k == "grad_norm" else 1e-5, atol=1e-6, err_msg=f"step {i} {k}")
        diff = (tr.flat_p - p_eager).abs()
        assert float(diff.max()) <= 2.1e-4 and float((diff > 2e-6).float().mean()) < 2e-3, f"step {i}"
    assert tr.cache_stats == {"captures": 3, "steps": 5}          # classes 2048 (x2), 3800 (x2), 1024


@gpu
def test_frozen_stage_prefetch_equals_the_in_line_step():
    """engine.Trainer's frozen-stage prefetch: announcing the next batch (`next_samples`) makes its stem + layer1 run beside the
    current step's matcher / backward; the training trajectory is the one of the trainer that runs them in line (same losses step by
    step, same parameters to atomic-order noise) -- over batches of changing shapes, an announced batch that does not come (another
    tensor arrives: in-line run), a batch modified after it was announced (version counter: in-line run), and un-announced steps."""
    from counting_detr_amd.engine import Trainer
    trs = []
    for on in (True, False):
        model, crit, args = _build(Q=100)
        args.frozen_prefetch = on
        trs.append(Trainer(model, crit, args, device=DEV))
    ta, tb = trs
    assert ta._prefetch_ok() and not tb._prefetch_ok()
    shapes = [(2, 64, 96), (2, 64, 96), (2, 96, 128), (2, 64, 96), (2, 64, 96), (2, 64, 96), (2, 96, 128)]
    batches = [_batch(B, H, W, (5 + i, 20 - i), seed=500 + i) for i, (B, H, W) in enumerate(shapes)]
    decoy = _batch(2, 64, 96, (3, 3), seed=999)[0]
    for i, (images, rects, tg) in enumerate(batches):
        for dst, src in zip((ta.flat_p, ta.exp_avg, ta.exp_avg_sq, ta.opt_state), (tb.flat_p, tb.exp_avg, tb.exp_avg_sq, tb.opt_state)):
            dst.copy_(src)                                # both trainers take every step from the same state: no drift to allow for
        nxt = batches[i + 1][0] if i + 1 < len(batches) else None
        if i == 3:
            nxt = decoy                                   # announced, never delivered
        if i == 5:
            nxt = None                                    # not announced
        oa = ta.step(images, rects, tg, next_samples=nxt)
        if i == 1:
            batches[2][0].add_(0.25)                      # the announced batch changes before it is delivered: its prefetched stage is stale
        ob = tb.step(images, rects, tg)
        torch.cuda.synchronize()
        for k in ob:
            np.testing.assert_allclose(float(oa[k]), float(ob[k]), rtol=2e-4 if k == "grad_norm" else 2e-5, atol=1e-6, err_msg=f"step {i} {k}")
    diff = (ta.flat_p - tb.flat_p).abs()
    assert float(diff.max()) <= 2.1e-4 and float((diff > 2e-6).float().mean()) < 2e-3
    # steps 1, 3 (same shape as announced) hit; 2 was modified, 4 got another tensor than announced, 6 was not announced; 0 is the first
    # step; every first meeting of a key (steps 0 and 2) recomputes in line inside the capture
    assert ta.prefetch_stats["hits"] == 3, ta.prefetch_stats          # steps 1, 3, 5
    assert tb.prefetch_stats == {"hits": 0, "inline": 0}


@gpu
def test_pipelined_replay_equals_plain_replay():
    """Trainer.replay(pipelined=True) (bench.py's fixed-batch loop: every step computes the frozen stage for the following one) ==
    Trainer.replay() step by step."""
    from counting_detr_amd.engine import Trainer
    outs = []
    for pipelined in (True, False):
        model, crit, args = _build(Q=100)
        tr = Trainer(model, crit, args, device=DEV)
        tr.capture(*_batch(2, 96, 128, (7, 30), seed=5), warmup=0)
        losses = []
        for _ in range(4):
            losses.append(float(tr.replay(pipelined=pipelined)["loss"]))
        torch.cuda.synchronize()
        outs.append((losses, tr.flat_p.detach().clone(), dict(tr.prefetch_stats)))
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=3e-3)          # (two runs drift apart by the atomic-order noise of their updates)
    assert outs[0][0][-1] < outs[0][0][0]                 # it trains
    assert outs[0][2] == {"hits": 3, "inline": 1} and outs[1][2] == {"hits": 0, "inline": 4}
    assert float((outs[0][1] - outs[1][1]).abs().max()) <= 1e-3


@gpu
def test_graph_cache_evicts_least_recently_used():
    from counting_detr_amd.engine import Trainer
    model, crit, args = _build(Q=100)
    args.graph_cache_size = 2
    tr = Trainer(model, crit, args, device=DEV)
    shapes = [(64, 96), (96, 96), (64, 96), (64, 64), (96, 96)]
    for i, (H, W) in enumerate(shapes):
        out = tr.step(*_batch(1, H, W, (5,), seed=i))
        assert np.isfinite(float(out["loss"]))
    # (64,96) (96,96) cached; (64,96) hit; (64,64) evicts (96,96); (96,96) captured again, evicting (64,96)
    assert tr.cache_stats["captures"] == 4 and len(tr._cache) == 2
    assert [k[0][2:] for k in tr._cache] == [(64, 64), (96, 96)]


def _write_fsc147(root, sizes, counts, seed=0):
    """An FSC-147-format training set (A2/data/fsc147.py:12-67): images_384_VarV2/*.png, annotation_FSC147_384.json with three
    exemplar boxes per image, annotations/pseudo_bbox_train.json with [cx, cy, w, h] pixel boxes."""
    from PIL import Image
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, "images_384_VarV2"))
    os.makedirs(os.path.join(root, "annotations"))
    anno, images, anns = {}, [], []
    aid = 1
    for i, ((h, w), n) in enumerate(zip(sizes, counts)):
        name = f"{i + 1}.png"
        Image.fromarray(rng.randint(0, 255, (h, w, 3), dtype=np.uint8)).save(os.path.join(root, "images_384_VarV2", name))
        ex = []
        for _ in range(3):
            x1, y1 = rng.uniform(0.1, 0.6) * w, rng.uniform(0.1, 0.6) * h
            x2, y2 = x1 + 0.1 * w, y1 + 0.15 * h
            ex.append([[x1, y1], [x1, y2], [x2, y2], [x2, y1]])
        anno[name] = {"box_examples_coordinates": ex}
        images.append({"id": i + 1, "file_name": name, "width": w, "height": h})
        for _ in range(n):
            anns.append({"id": aid, "image_id": i + 1, "category_id": 1, "iscrowd": 0,
                         "bbox": [float(rng.uniform(0.1, 0.9) * w), float(rng.uniform(0.1, 0.9) * h), float(rng.uniform(4, 12)), float(rng.uniform(4, 12))]})
            aid += 1
    json.dump(anno, open(os.path.join(root, "annotation_FSC147_384.json"), "w"))
    json.dump({"images": images, "annotations": anns, "categories": [{"id": 1, "name": "fg"}]},
              open(os.path.join(root, "annotations", "pseudo_bbox_train.json"), "w"))


@gpu
def test_main_trains_a_multi_size_dataset_through_the_graph_cache(tmp_path, capsys):
    """main.py (reader -> collate -> prefetch -> Trainer.step) on a generated FSC-147-format set with three image sizes and target
    counts from 0 to more than the number of queries: two epochs, every step after the first epoch is a cache hit, the result
    equals the run with --no_graph_cache (stream-ordered step) to the noise of the steps' atomics."""
    import main as main_mod
    from counting_detr_amd.args import get_args_parser
    root = str(tmp_path / "data")
    sizes = [(64, 100), (64, 100), (96, 130), (96, 130), (64, 70), (64, 70), (96, 130), (96, 130)]
    counts = [5, 9, 17, 0, 3, 3, 120, 2]
    _write_fsc147(root, sizes, counts)
    base = ["-dp", root, "--no_aux_loss", "--num_query_pattern", "1", "--num_query_position", "100", "--images_per_gpu", "2", "--device", DEV,
            "--epochs", "2", "--num_workers", "0", "--seed", "7"]
    losses = {}
    for tag, extra in (("graph", []), ("eager", ["--no_graph_cache"])):
        out = str(tmp_path / tag)
        main_mod.main(get_args_parser().parse_args(base + ["-o", out] + extra))
        lines = [json.loads(l) for l in open(os.path.join(out, "detr_retrain.txt")).read().strip().splitlines()]
        assert [l["epoch"] for l in lines] == [0, 1]
        losses[tag] = [l["train_loss"] for l in lines]
        if tag == "graph":
            assert lines[0]["train_graph_captures"] >= 1 and lines[1]["train_graph_steps"] == 4
    np.testing.assert_allclose(losses["graph"], losses["eager"], rtol=1e-2)


@gpu
def test_inference_engine_graph_equals_eager_forward():
    """engine.InferenceEngine (pre-split weight images + one captured graph per image shape) == the model's plain forward + counting
    rule on two shapes, the first shape seen twice (second time: a cache hit on other pixels / exemplars)."""
    from counting_detr_amd.engine import InferenceEngine, count_objects
    model, _, _ = _build(Q=100)
    eng = InferenceEngine(model)
    for i, (H, W) in enumerate([(96, 128), (64, 96), (96, 128)]):
        images, rects, _ = _batch(2, H, W, (1, 1), seed=40 + i)
        counts, keep, out, ref, prob = eng(images, rects)
        c0, k0, o0, r0 = count_objects(model, images, rects)
        assert torch.equal(counts, c0) and torch.equal(keep, k0)
        for k in ("pred_logits", "pred_boxes", "pred_vars"):
            np.testing.assert_allclose(out[k].cpu().numpy(), o0[k].cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=k)
        np.testing.assert_allclose(ref.cpu().numpy(), r0.cpu().numpy(), rtol=1e-6)
    assert eng.stats == {"captures": 2, "calls": 3, "prefetch_hits": 0}


@gpu
def test_inference_engine_prefetch_equals_plain_calls():
    """InferenceEngine with the next batch announced (its frozen stage runs on a side stream beside the current forward's encoder /
    decoder) returns bit-identical outputs to the un-announced calls -- same shape in a row, a shape change, an announced batch that is
    modified before it arrives (stale stage: recomputed in line) and one that never arrives."""
    from counting_detr_amd.engine import InferenceEngine
    model, _, _ = _build(Q=100)
    plain, piped = InferenceEngine(model, prefetch=False), InferenceEngine(model, prefetch=True)
    shapes = [(96, 128), (96, 128), (64, 96), (96, 128), (96, 128), (96, 128)]
    batches = [_batch(2, H, W, (1, 1), seed=60 + i) for i, (H, W) in enumerate(shapes)]
    decoy = _batch(2, 96, 128, (1, 1), seed=99)[0]
    for i, (images, rects, _) in enumerate(batches):
        nxt = batches[i + 1][0] if i + 1 < len(batches) else None
        if i == 3:
            nxt = decoy
        out_b = piped(images, rects, next_samples=nxt)[2]
        if i == 0:
            batches[1][0].mul_(1.5)                       # announced, then changed: the prefetched stage must not be used
        out_a = plain(images, rects)[2]
        for k in ("pred_logits", "pred_boxes", "pred_vars"):
            assert torch.equal(out_a[k], out_b[k]), f"call {i} {k}"
    assert piped.stats["prefetch_hits"] == 2              # calls 3 (announced by 2) and 5 (announced by 4); 1 was modified, 2 is a new shape's capture, 4 got another tensor
    assert plain.stats["prefetch_hits"] == 0


@gpu
def test_graph_layouts_agree(monkeypatch):
    """The captured step's layouts -- "chain" (default: linear graphs only, side work as separate graphs on side streams), "single" (round
    3: [forward] | [the rest with in-graph branches]) and "single" with the backbone's backward as three more sub-graphs (what
    world_size > 1 replayed in round 3) -- are the same step: same losses bit for bit on the first step (the forward and the matching
    are deterministic), same gradient norm and parameters to the atomic-order noise of the weight-gradient reductions."""
    from counting_detr_amd.engine import Trainer
    images, rects, tg = _batch(2, 128, 160, (7, 13), seed=21)
    res = []
    for layout, seg in (("single", "0"), ("single", "1"), ("chain", "0")):
        monkeypatch.setenv("CDETR_GRAPH_LAYOUT", layout)
        monkeypatch.setenv("CDETR_SEGMENTED_GRAPH", seg)
        model, crit, args = _build(Q=100)
        tr = Trainer(model, crit, args, device=DEV)
        tr.capture(images, rects, tg, warmup=0)
        e = tr._entry
        assert e["layout"] == layout
        if layout == "single":
            assert (e["segs"] is not None) == (seg == "1")
        else:
            assert len(e["S"]) == 3 and e["W"][0] is not None and e["W0"] is not None
            # the side streams were PROBED concurrent with the main stream and with each other (warm candidates: a cold stream's first
            # launch takes 0.2-40 ms and looks serialised): most of 8 candidates sit on another hardware queue than the main stream
            pr = tr.side_stream_probe
            assert pr["overlap_main"] >= 2 and pr["wg_overlaps_pf"], pr
        outs = []
        for _ in range(3):
            outs.append({k: float(v) for k, v in tr.replay().items()})
        torch.cuda.synchronize()
        res.append((outs, tr.flat_p.detach().clone()))
    (o0, p0) = res[0]
    for (o1, p1) in res[1:]:
        for k in o0[0]:
            if k != "grad_norm":
                assert o0[0][k] == o1[0][k], k                      # first step: identical weights -> identical forward, matching and losses
            for a, b in zip(o0, o1):
                np.testing.assert_allclose(b[k], a[k], rtol=2e-3, atol=1e-6, err_msg=k)
        diff = (p0 - p1).abs()
        assert float(diff.max()) <= 6.5e-4 and float((diff > 1e-5).float().mean()) < 1e-2      # 3 AdamW steps of <= lr each on near-zero gradients


@gpu
def test_cached_graph_step_with_aux_losses_equals_eager_step():
    """aux_loss=True (A2/models/anchor_detr.py:334-350: one Hungarian matching per decoder layer) through the graph cache: the layers are
    matched one after the other with the device-resident capacity plan; same 31 loss values as the stream-ordered step (whose matchings
    are one stacked launch), on two batches with different target counts replayed from ONE captured graph."""
    import counting_detr_amd
    from counting_detr_amd.args import default_args
    from counting_detr_amd.engine import Trainer
    from counting_detr_amd.init import seeded_init_
    args = default_args(device=DEV, num_query_position=100, aux_loss=True)
    model, crit, _ = counting_detr_amd.build_model(args)
    seeded_init_(model)
    model.to(DEV).train()
    tr = Trainer(model, crit, args, device=DEV)
    for i, Ts in enumerate([(7, 13), (40, 2)]):
        images, rects, tg = _batch(2, 96, 128, Ts, seed=60 + i)
        saved = [t.detach().clone() for t in (tr.flat_p, tr.exp_avg, tr.exp_avg_sq, tr.opt_state)]
        eo = {k: float(v) for k, v in tr.train_step(images, rects, tg).items()}
        for dst, src in zip((tr.flat_p, tr.exp_avg, tr.exp_avg_sq, tr.opt_state), saved):
            dst.copy_(src)
        go = {k: float(v) for k, v in tr.step(images, rects, tg).items()}
        assert len(eo) >= 31 and set(eo) == set(go)
        for k in eo:
            np.testing.assert_allclose(go[k], eo[k], rtol=1e-4 if k == "grad_norm" else 1e-5, atol=1e-6, err_msg=f"batch {i} {k}")
    assert tr.cache_stats == {"captures": 1, "steps": 2}


@gpu
def test_two_engines_of_different_arithmetic_in_one_process():
    """A split-bf16 Trainer and an fp32-MFMA InferenceEngine (another model) interleaved in one process: each computes in the arithmetic
    it was built with (ops.arithmetic scopes; the module defaults are untouched) -- the evaluator's outputs are bit-identical to the ones
    it produces alone, the trainer's first-step losses too."""
    from counting_detr_amd import ops
    from counting_detr_amd.engine import InferenceEngine, Trainer
    default = (ops.PRECISION, ops.PRECISION_BWD)
    images, rects, tg = _batch(2, 96, 128, (7, 13), seed=77)

    def build_pair():
        model, crit, args = _build(Q=100)
        tr = Trainer(model, crit, args, device=DEV, precision=1, precision_bwd=3)
        model2, _, _ = _build(Q=100)
        return tr, InferenceEngine(model2, precision=0)
    tr, ev = build_pair()
    alone_ev = [t.clone() for t in (ev(images, rects)[2]["pred_logits"], ev(images, rects)[2]["pred_boxes"])]
    tr2, ev2 = build_pair()
    alone_tr = {k: float(v) for k, v in tr2.step(images, rects, tg).items()}
    tr3, ev3 = build_pair()
    assert tr3.arith == (1, 3) and ev3.arith[0] == 0
    o1 = ev3(images, rects)[2]
    mixed_ev = [o1["pred_logits"].clone(), o1["pred_boxes"].clone()]
    mixed_tr = {k: float(v) for k, v in tr3.step(images, rects, tg).items()}
    o2 = ev3(images, rects)[2]
    for a, b, c in zip(alone_ev, mixed_ev, (o2["pred_logits"], o2["pred_boxes"])):
        assert torch.equal(a, b) and torch.equal(a, c)
    for k, v in alone_tr.items():
        if k != "grad_norm":
            assert mixed_tr[k] == v, k
    assert (ops.PRECISION, ops.PRECISION_BWD) == default
    # and the two arithmetics really differ: the fp32-MFMA evaluator of the TRAINER's weights is not bit-equal to a split-bf16 one
    e_a, e_b = InferenceEngine(tr3.model, precision=0), InferenceEngine(tr3.model, precision=1)
    assert not torch.equal(e_a(images, rects)[2]["pred_boxes"], e_b(images, rects)[2]["pred_boxes"])
