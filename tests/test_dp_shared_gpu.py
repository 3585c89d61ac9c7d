"""Data-parallel step on the real HIP kernels, two ranks sharing ONE GPU (gloo carries the collectives through the host):
the bucketed exchange fired from the backbone's backward hooks, the deferred parameter-gradient flush before each bucket,
the global loss normaliser and the two-graph replay must reproduce the single-process step on the concatenated batch
(A2/main.py:137-142 wraps the same model in DistributedDataParallel; A2/models/anchor_detr.py:321-325 normaliser)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

H = W = 256
Q = 100
TS = (5, 9, 3, 7)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(dev):
    import counting_detr_amd
    from counting_detr_amd.args import default_args
    from counting_detr_amd.engine import Trainer
    from counting_detr_amd.init import seeded_init_
    args = default_args(device=str(dev), num_query_position=Q)
    model, crit, _ = counting_detr_amd.build_model(args)
    seeded_init_(model)
    model.to(dev).train()
    crit.train()
    return Trainer(model, crit, args, device=dev)


def _batch(dev):
    from bench import synthetic_batch
    return synthetic_batch(4, H, W, TS, seed=7, device=dev)


def _worker(rank, world, port, use_graph, precision, ret):
    from counting_detr_amd import ops
    ops.PRECISION = precision
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        tr = _make(dev)
        images, rects, targets = _batch(dev)
        lo, hi = 2 * rank, 2 * rank + 2
        shard = (images[lo:hi].contiguous(), rects[lo:hi].contiguous(), targets[lo:hi])
        if use_graph:
            tr.capture(*shard)
            out = tr.replay()
        else:
            out = tr.train_step(*shard)
        torch.cuda.synchronize()
        ret[rank] = {"g": tr.flat_g.detach().cpu(), "p": tr.flat_p.detach().cpu(), "loss": float(out["loss"]),
                     "gn": float(out["grad_norm"]), "on_side": bool(tr.exchange.on_side)}
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_graph,precision", [(False, 0), (True, 0), (False, 1), (True, 1)])
def test_two_ranks_match_single_process_on_the_concatenated_batch(use_graph, precision, monkeypatch):
    from counting_detr_amd import ops
    # The fp32 cases pin the exchange mechanics to 1e-4: that needs the FORWARD of an image to be bit-identical in a batch of two and
    # of four, which holds as long as a GEMM's summation order does not depend on its row count.  The split reductions
    # (cdetr_gemm_desc.splitk_ws) choose their slice count from the grid size, i.e. from the batch: activations then differ in the
    # last bits (measured 2e-6 in round 2: tools/sk_check3.py, in the git history since the round-6 pruning) and this random-init network turns that into ~1e-3 on the gradient (the same
    # ~500x it shows for the 1.5e-5 of the bf16x3 rounding, matching unchanged -- round 2, tools/sk_check.py, git history).  So: split off where the
    # tolerance is the fp32 one, on (the default) in the bf16x3 cases.
    if precision == 0:
        monkeypatch.setenv("CDETR_SPLITK", "0")          # the spawned ranks read it at import
        monkeypatch.setattr(ops, "SPLITK", 0)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), use_graph, precision, ret), nprocs=2, join=True)
    old, ops.PRECISION = ops.PRECISION, precision
    try:
        _compare(ret)
    finally:
        ops.PRECISION = old


def test_two_ranks_with_the_buckets_issued_from_the_side_stream(monkeypatch):
    """FlatGradExchange.on_side (CDETR_EXCHANGE_ON_SIDE=1, round 6): the buckets are issued asynchronously from the weight-gradient stream and
    the compute stream waits for the work handles in finish() -- the captured chain replay must give the same reduced gradient, the same
    parameters on both ranks and the same agreement with the single-process step as the default form."""
    from counting_detr_amd import ops
    monkeypatch.setenv("CDETR_EXCHANGE_ON_SIDE", "1")        # the spawned ranks read it when their FlatGradExchange is built
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), True, 1, ret), nprocs=2, join=True)
    assert all(ret[r].get("on_side") for r in (0, 1)), "the ranks did not run the on-side form"
    old, ops.PRECISION = ops.PRECISION, 1
    try:
        _compare(ret)
    finally:
        ops.PRECISION = old


def _compare(ret):
    r0, r1 = ret[0], ret[1]
    assert torch.equal(r0["g"], r1["g"]), "ranks disagree on the reduced gradient"
    assert torch.equal(r0["p"], r1["p"]), "ranks diverged after one step"
    dev = torch.device("cuda", 0)
    tr = _make(dev)
    images, rects, targets = _batch(dev)
    out = tr.train_step(images, rects, targets)
    torch.cuda.synchronize()
    g1 = tr.flat_g.detach().cpu()
    g2 = r0["g"] / 2                      # the exchange sums; the 1/world average is applied inside the AdamW kernel
    scale = g1.abs().max().item()
    assert scale > 0
    err = (g1 - g2).abs().max().item() / scale
    rel2 = ((g1 - g2).norm() / g1.norm()).item()
    print(f"dp vs single: max err / max |g| = {err:.2e}, l2 relative = {rel2:.2e}")
    # the two runs tile their GEMMs differently (B=2 vs B=4 grids): fp32 products agree to rounding, bf16x3 to ~2^-16 per
    # product amplified through 100+ layers of a random-init network
    from counting_detr_amd import ops
    lim = 1e-4 if ops.PRECISION == 0 else 5e-3
    assert err < lim and rel2 < lim, f"data-parallel gradient differs from the single-process one: {err:.2e} / {rel2:.2e}"
    gn1 = float(out["grad_norm"])
    assert abs(r0["gn"] - gn1) <= 1e-3 * gn1
    # per-rank losses are each normalised by (global boxes / world): their mean is the single-process loss
    assert abs((r0["loss"] + r1["loss"]) / 2 - float(out["loss"])) <= 1e-3 * abs(float(out["loss"]))


def _cache_worker(rank, world, port, ret):
    """Trainer.step (graph cache, five sub-graphs + bucketed exchange between them) on a schedule where the two ranks meet image sizes in
    DIFFERENT orders: at most steps one rank captures a new key while the other replays a cached one and sits in its all-reduce."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bench import synthetic_batch
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        tr = _make(dev)
        sizes = [(128, 160), (96, 128), (128, 160), (64, 96), (96, 128)] if rank == 0 else [(96, 128), (96, 128), (64, 96), (128, 160), (96, 128)]
        losses = []
        for i, (h, w) in enumerate(sizes):
            images, rects, targets = synthetic_batch(2, h, w, (3 + i, 9 - i), seed=10 * rank + i, device=dev)
            losses.append(float(tr.step(images, rects, targets)["loss"]))
        torch.cuda.synchronize()
        ret[rank] = {"p": tr.flat_p.detach().cpu(), "losses": losses, "stats": dict(tr.cache_stats)}
    finally:
        dist.destroy_process_group()


def test_graph_cache_with_ranks_meeting_shapes_in_different_orders():
    """Data-parallel Trainer.step: no rendezvous is needed to capture (capture_error_mode="thread_local", collectives only BETWEEN the
    sub-graphs), so ranks that meet new shapes at different steps neither deadlock nor diverge: bit-identical parameters after 5 steps."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_cache_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    r0, r1 = ret[0], ret[1]
    assert torch.equal(r0["p"], r1["p"]), "replicas diverged"
    assert r0["stats"] == {"captures": 3, "steps": 5} and r1["stats"] == {"captures": 3, "steps": 5}
    assert all(l == l and abs(l) < 1e4 for l in r0["losses"] + r1["losses"])


@pytest.mark.gpu
def test_bucket_allreduces_can_be_captured_by_rccl(tmp_path):
    """FlatGradExchange.capture_buckets (the selectable captured form of the gradient exchange, --captured_allreduce): RCCL's all-reduce is
    recorded into HIP graphs on the exchange stream and replayed -- on a ONE-rank NCCL group (the only kind a one-GPU box can form: the sum
    over one rank is the identity), which exercises the capture / replay mechanics, not the wire."""
    import subprocess
    import sys
    script = tmp_path / "cap.py"
    script.write_text('''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29711")
dist.init_process_group("nccl", rank=0, world_size=1)
from counting_detr_amd.engine import FlatGradExchange
g = torch.arange(1000, dtype=torch.float32, device="cuda")
ex = FlatGradExchange(g, [0, 400, 400, 900, 1000])
dist.all_reduce(g)                       # communicator set up outside any capture
torch.cuda.synchronize()
assert ex.capture_buckets() and [x is not None for x in ex.graphs] == [True, False, True, True]
ref = g.clone()
for x in ex.graphs:
    if x is not None:
        with torch.cuda.stream(ex.stream):
            x.replay()
torch.cuda.synchronize()
assert torch.equal(g, ref)
print("captured-ok")
dist.destroy_process_group()
''' % ROOT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert "captured-ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
