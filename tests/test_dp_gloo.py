"""world_size = 2 (gloo, CPU) tests of the data-parallel plumbing: bucketed gradient exchange, loss normaliser and
logging reductions.  The model kernels need a GPU, so these tests drive the communication layer with CPU tensors."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def run2(fn):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), fn, ret), nprocs=2, join=True)
    return dict(ret)


def _exchange(rank, world):
    from counting_detr_amd.engine import FlatGradExchange
    g = torch.arange(100, dtype=torch.float32) * (rank + 1)
    ex = FlatGradExchange(g, [0, 40, 70, 90, 100])
    for seg in (0, 1, 2, 3):          # the order backward announces them
        ex.segment_done(seg)
    ex.finish()
    return g.tolist()


def test_bucketed_allreduce_sums_every_segment():
    out = run2(_exchange)
    ref = (torch.arange(100, dtype=torch.float32) * 3).tolist()
    assert out[0] == ref and out[1] == ref


def _exchange_missing(rank, world):
    from counting_detr_amd.engine import FlatGradExchange
    g = torch.ones(10) * (rank + 1)
    ex = FlatGradExchange(g, [0, 4, 4, 8, 10])
    ex.segment_done(0)                # segments 1 (empty), 2, 3 never announced -> finish() reduces them
    ex.finish()
    return g.tolist()


def test_finish_reduces_unannounced_segments():
    out = run2(_exchange_missing)
    assert out[0] == [3.0] * 10 and out[1] == [3.0] * 10


def _num_boxes(rank, world):
    from counting_detr_amd.misc import get_world_size, reduce_dict
    # the loss normaliser of A2/models/anchor_detr.py:321-325: sum over ranks / world, clamped at 1
    nb = torch.tensor([37.0 + 120.0] if rank == 0 else [0.0])
    dist.all_reduce(nb)
    nb = torch.clamp(nb / get_world_size(), min=1)
    red = reduce_dict({"loss_ce": torch.tensor(float(rank + 1)), "loss_bbox": torch.tensor(2.0 * (rank + 1))})
    return float(nb), {k: float(v) for k, v in red.items()}


def test_num_boxes_and_reduce_dict():
    out = run2(_num_boxes)
    for r in (0, 1):
        assert out[r][0] == pytest.approx(78.5)
        assert out[r][1] == {"loss_bbox": pytest.approx(3.0), "loss_ce": pytest.approx(1.5)}


def _equiv(rank, world):
    """DP convention check: SUM of per-rank grads scaled by 1/world (folded into the optimizer) equals the gradient of
    the single-process large batch when every rank normalises its loss by num_boxes/world."""
    torch.manual_seed(0)
    w = torch.randn(5, requires_grad=True)
    xs = torch.randn(4, 5)
    t_counts = [3.0, 5.0]
    num_boxes = sum(t_counts) / world
    local = xs[rank * 2:(rank + 1) * 2]
    loss = (local @ w).pow(2).sum() / num_boxes
    loss.backward()
    g = w.grad.clone()
    dist.all_reduce(g)
    g /= world
    w2 = w.detach().clone().requires_grad_(True)
    ((xs @ w2).pow(2).sum() / sum(t_counts)).backward()
    return torch.allclose(g, w2.grad, rtol=1e-5, atol=1e-6)


def test_dp_gradient_equals_large_batch():
    out = run2(_equiv)
    assert out[0] and out[1]


def _replicas(rank, world):
    """Ranks that build their model from DIFFERENT RNG streams (what main.py used to do) hold identical parameters, frozen
    weights and buffers once the trainer exists (DistributedDataParallel's construction-time broadcast, A1/main.py:206-208)."""
    import hashlib
    from counting_detr_amd import build_model
    from counting_detr_amd.args import default_args
    from counting_detr_amd.engine import Trainer
    args = default_args()
    args.device = "cpu"
    torch.manual_seed(1234 + rank)
    model, crit, _ = build_model(args)
    with torch.no_grad():
        model.backbone.body.bn1.running_mean.add_(float(rank))          # a buffer and a frozen weight that differ per rank
        model.backbone.body.layer1[0].conv1.weight.add_(float(rank))
    tr = Trainer(model, crit, args, device="cpu")
    h = hashlib.sha256()
    h.update(tr.flat_p.numpy().tobytes())
    for k, v in sorted(model.state_dict().items()):
        h.update(v.detach().contiguous().numpy().tobytes())
    s1, b1 = model.backbone.body.bn1.affine()
    return h.hexdigest(), float(model.backbone.body.bn1.running_mean[0]), float(b1[0])


def test_trainer_broadcasts_rank0_weights_to_every_replica():
    out = run2(_replicas)
    assert out[0][0] == out[1][0], "replicas start from different weights"
    assert out[0][1] == out[1][1] and out[0][2] == out[1][2]
