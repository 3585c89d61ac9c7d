"""bench.py -- images/sec of one FSCD-147 2nd-stage Counting-DETR training step on N MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps 20 --warmup 5          (N > 1: re-launches itself as N ranks, one per GPU, over RCCL)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = forward + device Hungarian matcher + SetCriterion + backward + clip_grad_norm(0.1) + AdamW on a synthetic
batch of 2 images 800x800 per GPU (Q=300 learned anchors, T=(37,120) targets -- BASELINE.json configs[1]),
inputs resident in HBM, random-init (name-seeded) weights.  Weak scaling: per-GPU batch fixed, gradients averaged over
ranks by RCCL in four buckets that leave while the backbone's backward is still running (stream-ordered step: from hooks in
the backward; graph replay: between the five captured sub-graphs).
Prints ONE JSON line on rank 0 with the extra objects `roofline` (implicit-GEMM family timed with HIP events on the launch
stream in an instrumented eager pass of the same step), `step_ms` (median / p10 / p90 of the timed steps, HIP events),
`extra_shapes` (the shipped script's grid-576 shape and a 384x576 image), at N>1 `allreduce_exposed_ms`, and at N=1
`cpu_baseline` (the oracle's CPU restatement of the same step on the host's physical cores, bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0       # dense bf16 MFMA; the split-bf16 mode spends 3 bf16 MFMAs per algorithmic product
PRECISIONS = {"bf16x3": 1, "fp32": 0}
TRAFFIC_JSON = next((p for p in (os.path.join(ROOT, "profiles", f"r{r}_traffic.json") for r in (6, 5, 4, 3, 2)) if os.path.exists(p)), "")


def step_gflop_per_image(H, W, Q):
    """algorithmic work of one train step per image, reduced form (mean-before-project keys):
    616 GFLOP at 800x800 / Q=300, 643 at Q=576, 212 at 384x576 / Q=300."""
    C, nh, d, dff = 256, 8, 32, 1024
    px, h, w = H * W, (H + 15) // 16, (W + 15) // 16      # three stride-2 stages + the max-pool, each rounding up
    n = h * w
    backbone = 123697.0 * px
    proj = 4096.0 * 256 * n
    lo, hi = min(h, w), max(h, w)
    enc = n * C * C * 6 + nh * n * (w + h) * d + nh * n * lo * hi * d + nh * n * hi * d + 2 * n * C * dff
    dec = (4 * Q * C * C + 2 * nh * Q * Q * d) + (2 * Q * C * C + 3 * n * C * C + nh * Q * (w + h) * d + nh * Q * lo * hi * d
                                                  + nh * Q * hi * d + Q * C * C) + 6 * Q * C * C + Q * (4 * C * C + 8 * C) + 2 * Q * C * dff
    fwd = backbone + proj + 6 * enc + 6 * dec - 12 * 2 * (n - (w + h) / 2.0) * C * C
    frozen = 10.02e9 * px / 640000.0
    return 2.0 * (fwd + 2.0 * (fwd - frozen)) / 1e9


def dtype_string(precision):
    """The arithmetic the step computes in, pass by pass (never a narrower claim than what runs)."""
    if precision == "fp32":
        return "fp32 (fp32 MFMA, exact products; fp32 storage)"
    from counting_detr_amd import ops
    bwd = {1: "bf16x3", 2: "bf16x2 (weight / activation operand rounded to bf16, incoming gradient split hi+lo, 2 MFMAs per product)",
           3: "bf16 (both operands rounded to bf16, 1 MFMA per product)"}[ops.PRECISION_BWD]
    return ("forward bf16x3 (fp32 operands split hi+lo, 3 bf16 MFMAs per product); data / weight gradients " + bwd +
            "; fp32 accumulate, fp32 storage, fp32 attention softmax / norms / losses / optimizer")


def synthetic_batch(B, H, W, Ts, seed, device):
    g0 = torch.Generator().manual_seed(seed)
    images = torch.randn(B, 3, H, W, generator=g0)
    rects = torch.tensor([[.10, .10, .20, .20], [.40, .40, .50, .55], [.70, .20, .80, .30]])[None].repeat(B, 1, 1)
    g1 = torch.Generator().manual_seed(seed + 1)
    targets = []
    for b in range(B):
        T = Ts[b % len(Ts)]
        cxcy = torch.rand(T, 2, generator=g1) * 0.8 + 0.1
        wh = torch.rand(T, 2, generator=g1) * 0.10 + 0.02
        targets.append({"boxes": torch.cat([cxcy, wh], 1).to(device), "labels": torch.zeros(T, dtype=torch.int64, device=device)})
    return images.to(device), rects.to(device), targets


def physical_cores():
    """Physical cores of the host (unique (physical id, core id) pairs of /proc/cpuinfo), not os.cpu_count()'s SMT threads."""
    try:
        pairs, phys, core = set(), None, None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        pairs.add((phys, core))
                    phys = core = None
        if pairs:
            n = len(pairs)
            try:
                n = min(n, len(os.sched_getaffinity(0)))      # a container may pin fewer CPUs than the host has
            except Exception:
                pass
            return max(n, 1)
    except Exception:
        pass
    return max((os.cpu_count() or 2) // 2, 1)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(B, H, W, Ts, steps=3, warmup=2):
    """The oracle (CPU restatement of the reference step, parity-pinned against the real reference) on the host cores:
    one step per candidate thread count (8 / 16 / 32 / 64 / all physical cores), then `warmup` + `steps` timed steps at the best
    one (>= 3 timed steps after 2 warm-ups, physical cores)."""
    from oracle.step import OracleTrainer, synthetic_batch as sb
    phys = physical_cores()
    tr = OracleTrainer(num_position=300)
    images, rects, targets = sb(B=B, H=H, W=W, Ts=Ts)
    cands = sorted({c for c in (8, 16, 32, 64, phys) if c <= phys} or {phys})
    torch.set_num_threads(cands[-1])
    tr.step(images, rects, targets)                      # first touch: allocator / oneDNN primitive caches
    sweep = {}
    for c in cands:
        torch.set_num_threads(c)
        t0 = time.perf_counter()
        tr.step(images, rects, targets)
        sweep[c] = time.perf_counter() - t0
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    for _ in range(max(warmup - 1, 0)):                  # the sweep step at `best` already was a warm-up at this thread count
        tr.step(images, rects, targets)
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        tr.step(images, rects, targets)
        ts.append(time.perf_counter() - t0)
    dt = sum(ts) / len(ts)
    return {"value": B / dt, "unit": "images/s", "cores": best, "kind": "port", "physical_cores": phys, "cpu": cpu_model(),
            "sample": f"{steps} timed steps (+{warmup} warm-ups) of the same B={B} {H}x{W} Q=300 T={list(Ts)} step, oracle fp32 on "
                      f"CPU at the best of {cands} threads",
            "ms_per_step": dt * 1e3, "thread_sweep_ms": {str(k): v * 1e3 for k, v in sweep.items()}}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: become the launcher -- N ranks of this same command line under
    torch.distributed.run, one per GPU, rendezvous on 127.0.0.1.  Output and exit code are the ranks'."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("GPU_MAX_HW_QUEUES", "8")      # main + prefetch + weight-gradient + exchange streams + RCCL's own: more than HIP's default 4 queues
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def percentiles(xs):
    xs = sorted(xs)
    if not xs:
        return None
    pick = lambda q: xs[min(len(xs) - 1, max(0, int(round(q * (len(xs) - 1)))))]     # noqa: E731
    return {"median": pick(0.5), "p10": pick(0.1), "p90": pick(0.9), "min": xs[0], "max": xs[-1], "n": len(xs)}


def timed_steps(step, n, barrier):
    """EXACTLY n steps between two barrier + device-sync brackets (host clock), plus one HIP event per step boundary for the
    per-step distribution (recording an event does not synchronise anything)."""
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    barrier()
    t0 = time.perf_counter()
    evs[0].record()
    out = None
    for i in range(n):
        out = step()
        evs[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    return dt, [evs[i].elapsed_time(evs[i + 1]) for i in range(n)], out


def exchange_variants(trainer, step, barrier, dev, share, rank, n=5):
    """First N > 1 contact, self-diagnosing: time `n` steps of each form of the gradient exchange -- {host-issued
    all-reduce, captured bucket graphs} x {exchange stream of its own, buckets issued from the weight-gradient stream
    (FlatGradExchange.on_side)} -- and keep the fastest for the contract's timed run.  Every rank takes the same decision (MAX over ranks
    per variant); a line per variant goes to stderr AS IT COMPLETES, so a variant that hangs names itself.  The default form is timed
    first.  By default the two host-issued forms; CDETR_BENCH_EXCHANGE_AB=all adds the captured ones (RCCL only), =0 skips all of this."""
    ex = trainer.exchange
    have_graphs = False
    # captured bucket graphs replay an RCCL collective that was never executed across two GPUs in this repository's history: they join the A/B
    # only on request (CDETR_BENCH_EXCHANGE_AB=all), so that the default N > 1 run -- the first one on real hardware -- carries nothing but
    # ordinary all-reduce calls (blocking and async_op) and cannot hang in a form nobody asked for
    if not share and os.environ.get("CDETR_BENCH_EXCHANGE_AB", "1") == "all":
        ok = 1.0
        try:
            if ex.graphs is None:
                ex.capture_buckets()
        except Exception as e_:                                # noqa: BLE001 -- a capture problem must not cost the run
            print(f"[bench] rank {rank}: capturing the bucket all-reduces failed ({type(e_).__name__}: {e_}); captured variants skipped", file=sys.stderr, flush=True)
            ex.graphs = None
            ok = 0.0
        okt = torch.tensor([ok if ex.graphs is not None else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        have_graphs = float(okt[0]) > 0
        if not have_graphs:
            ex.graphs = None
    variants = [("host_issued/own_stream", False, False)]
    if have_graphs:
        variants.append(("captured/own_stream", True, False))
    variants.append(("host_issued/side_stream", False, True))
    if have_graphs:
        variants.append(("captured/side_stream", True, True))
    res = {}
    for name, graphs, side in variants:
        ex.use_graphs, ex.on_side = graphs, side
        for _ in range(2):
            step()
        barrier()
        ex.probe = []
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        barrier()
        dt = (time.perf_counter() - t0) / n * 1e3
        exposed = ex.exposed_ms() or []
        ex.probe = None
        t = torch.tensor([dt, sum(exposed) / max(len(exposed), 1)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        res[name] = {"ms_per_step": float(t[0]), "allreduce_exposed_ms": float(t[1])}
        if rank == 0:
            print(f"[bench] exchange variant {name}: {float(t[0]):.3f} ms/step, exposed {float(t[1]):.3f} ms", file=sys.stderr, flush=True)
    best = min(res, key=lambda k: res[k]["ms_per_step"])
    _, graphs, side = next(v for v in variants if v[0] == best)
    ex.use_graphs, ex.on_side = graphs, side
    return {"variants": res, "kept": best, "steps_each": n, "probe": getattr(trainer, "side_stream_probe", None),
            "note": "ms_per_step = MAX over ranks of the mean of %d captured steps; the timed run uses `kept`" % n}


def build_trainer(dev, queries, prior, precision):
    import counting_detr_amd
    from counting_detr_amd import ops
    from counting_detr_amd.args import default_args
    from counting_detr_amd.engine import Trainer
    from counting_detr_amd.init import seeded_init_
    args = default_args(device=str(dev), num_query_position=queries, spatial_prior=prior)
    model, crit, _ = counting_detr_amd.build_model(args)
    seeded_init_(model)          # deterministic name-seeded random weights (no checkpoints in this environment)
    model.to(dev).train()
    crit.train()
    return Trainer(model, crit, args, device=dev, precision=PRECISIONS[precision])      # the trainer owns its arithmetic: no module global is flipped


def extra_shape(dev, H, W, queries, prior, Ts, batch, precision, steps=10):
    """A short graph-replay run of another shape (single GPU): capture, 3 warm-up replays, `steps` timed."""
    tr = build_trainer(dev, queries, prior, precision)
    images, rects, targets = synthetic_batch(batch, H, W, Ts, seed=0, device=dev)
    tr.capture(images, rects, targets, warmup=1)
    rp = (lambda: tr.replay(pipelined=True)) if tr._entry.get("fs") is not None else tr.replay      # one batch of look-ahead, like the main line
    for _ in range(3):
        rp()

    def bar():
        torch.cuda.synchronize()
    dt, per, out = timed_steps(rp, steps, bar)
    q = out["loss"]
    Q = tr.model.transformer.num_position * tr.model.transformer.num_pattern
    return {"image": [H, W], "queries": Q, "spatial_prior": prior, "targets": list(Ts), "images_per_gpu": batch,
            "value": batch * steps / dt, "unit": "images/s", "ms_per_step": dt / steps * 1e3, "step_ms": percentiles(per),
            "steps": steps, "whole_step_tflops": step_gflop_per_image(H, W, Q) * batch / (dt / steps * 1e3), "final_loss": float(q)}


def inference_leg(dev, shapes, batch0, precision, steps=20):
    """Forward + counting rule (A2/infer.py:57-81) through engine.InferenceEngine: pre-split weight images built once, one captured
    HIP graph per shape, replayed `steps` times after 3 warm-ups.  images/s per shape, plus the eager (stream-ordered) rate."""
    import counting_detr_amd
    from counting_detr_amd import ops
    from counting_detr_amd.args import default_args
    from counting_detr_amd.engine import InferenceEngine
    from counting_detr_amd.init import seeded_init_
    args = default_args(device=str(dev))
    model, _, _ = counting_detr_amd.build_model(args)
    seeded_init_(model)
    model.to(dev).eval()
    out = []
    for shp in shapes:
        (H, W), batch = shp[:2], (shp[2] if len(shp) > 2 else batch0)       # a third entry overrides the images per launch
        images, rects, _ = synthetic_batch(batch, H, W, (1,), seed=7, device=dev)
        row = {"image": [H, W], "images_per_gpu": batch, "queries": 300}
        for tag, graphs in (("graph", True), ("eager", False)):
            eng = InferenceEngine(model, graphs=graphs, precision=PRECISIONS[precision])
            # (graph mode, one batch of look-ahead like infer.py's loop: the next batch's frozen stage runs beside this batch's encoder / decoder)
            for _ in range(3):
                counts = eng(images, rects, next_samples=images)[0]
            torch.cuda.synchronize()
            dt, per, _ = timed_steps(lambda: eng(images, rects, next_samples=images), steps, torch.cuda.synchronize)
            row[tag] = {"value": batch * steps / dt, "unit": "images/s", "ms_per_batch": dt / steps * 1e3, "step_ms": percentiles(per)}
        row["counts"] = [int(c) for c in counts]
        out.append(row)
    return {"what": "forward + counting rule (sigmoid(logit[...,0]) >= 0.5), inputs resident in HBM, graph replay vs stream-ordered launches",
            "dtype": "forward bf16x3, fp32 accumulate / storage", "steps": steps, "shapes": out}


def stage1_leg(dev, precision, points=900, H=800, W=800, steps=20):
    """BASELINE config 5: 1st-stage pseudo-label generation (point -> box, A1/engine.py:124-187) with 900 anchor points on one 800x800
    image: forward of the stage-1 model (input_proj, encoder, decoder with 900 queries, wh head), graph replay and stream-ordered."""
    from counting_detr_amd import ops, stage1
    from counting_detr_amd.args import default_args
    from counting_detr_amd.engine import build_weight_mirror
    from counting_detr_amd.init import seeded_init_
    args = default_args(device=str(dev), spatial_prior="defined")
    model, _, _ = stage1.build(args)
    seeded_init_(model)
    model.to(dev).eval()
    mirror = build_weight_mirror(model, [(n, p) for n, p in model.named_parameters()], dgrad=False)
    mirror.refresh("fwd")
    g0 = torch.Generator().manual_seed(5)
    image = torch.randn(1, 3, H, W, generator=g0).to(dev)
    pts = (torch.rand(1, points, 2, generator=g0) * 0.9 + 0.05).to(dev)
    with ops.arithmetic(PRECISIONS[precision]), ops.scope(MIRROR=mirror):
        with torch.no_grad():
            run = lambda: stage1.generate_pseudo_boxes(model, image, pts)      # noqa: E731
            for _ in range(3):
                boxes = run()
            torch.cuda.synchronize()
            dt_e, per_e, _ = timed_steps(run, steps, torch.cuda.synchronize)
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                boxes = run()
            for _ in range(3):
                gr.replay()
            torch.cuda.synchronize()
            dt_g, per_g, _ = timed_steps(gr.replay, steps, torch.cuda.synchronize)
    return {"what": f"stage-1 forward (point -> box): {points} anchor points, one {H}x{W} image, pseudo boxes out", "steps": steps,
            "graph": {"value": steps / dt_g, "unit": "images/s", "ms_per_image": dt_g / steps * 1e3, "step_ms": percentiles(per_g)},
            "eager": {"value": steps / dt_e, "unit": "images/s", "ms_per_image": dt_e / steps * 1e3},
            "mean_box_wh": [float(boxes[..., 2].mean()), float(boxes[..., 3].mean())]}


def real_data_leg(dev, precision, batch=2, reps=4):
    """The loop main.py runs (engine.train_one_epoch -> Trainer.step: cached HIP graphs keyed by padded image size and target-capacity
    class) on batches shaped like FSC-147 after the reference's resize rule (384 high, widths multiples of 32, A2/data/fsc147.py:75-77)
    with target counts that change every step; ms/step of the steady state per image size next to the fixed-shape replay of the same
    size (Trainer.capture / replay, what `value` times).  In-memory batches: the reader / PIL decode is not part of the step."""
    tr = build_trainer(dev, 300, "learned", precision)
    sizes = [(384, 576), (384, 512), (384, 640)]
    counts = [(37, 120), (7, 64), (101, 3), (56, 0), (12, 128), (90, 77)]
    batches = []
    for r in range(reps * len(counts)):
        H, W = sizes[r % len(sizes)]
        Ts = counts[r % len(counts)] if r % 7 else (180, 20)                      # every 7th batch falls into the next capacity class (128 < T <= Q)
        images, rects, targets = synthetic_batch(batch, H, W, Ts, seed=300 + r, device=dev)
        batches.append(((H, W), Ts, images, rects, targets))
    nxt = lambda i: batches[i + 1][2] if i + 1 < len(batches) else None      # noqa: E731  (the look-ahead engine.train_one_epoch does)
    for i, b in enumerate(batches):                                              # first meeting of every key: captures
        tr.step(b[2], b[3], b[4], next_samples=nxt(i))
    torch.cuda.synchronize()
    cap0 = dict(tr.cache_stats)
    per_size = {}
    evs = []
    for i, b in enumerate(batches):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = tr.step(b[2], b[3], b[4], next_samples=nxt(i))
        e1.record()
        evs.append((b[0], e0, e1))
    t0 = time.perf_counter()
    for i, b in enumerate(batches):
        out = tr.step(b[2], b[3], b[4], next_samples=nxt(i))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / len(batches) * 1e3
    for sz, e0, e1 in evs:
        per_size.setdefault(sz, []).append(e0.elapsed_time(e1))
    rows = []
    for (H, W) in sizes:
        images, rects, targets = synthetic_batch(batch, H, W, (37, 120), seed=1, device=dev)
        tr.capture(images, rects, targets, warmup=0)
        rp = (lambda: tr.replay(pipelined=True)) if tr._entry.get("fs") is not None else tr.replay
        for _ in range(3):
            rp()
        dt, per, _ = timed_steps(rp, 10, torch.cuda.synchronize)
        fixed = dt / 10 * 1e3
        cached = sorted(per_size[(H, W)])[len(per_size[(H, W)]) // 2]
        rows.append({"image": [H, W], "cached_step_ms_median": cached, "fixed_replay_ms": fixed, "ratio": cached / fixed,
                     "steps": len(per_size[(H, W)])})
    return {"what": "Trainer.step over batches of 3 image sizes x 7 target-count tuples (2 capacity classes), steady state after the captures",
            "wall_ms_per_step_all_sizes": wall, "new_captures_in_timed_part": tr.cache_stats["captures"] - cap0["captures"],
            "graphs_cached": len(tr._cache), "per_size": rows, "final_loss": float(out["loss"]), "frozen_stage_prefetch": dict(tr.prefetch_stats),
            "device_memory_reserved_GB": torch.cuda.memory_reserved() / 2 ** 30}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, nargs=2, default=[800, 800])
    ap.add_argument("--batch", type=int, default=2, help="images per GPU")
    ap.add_argument("--queries", type=int, default=300)
    ap.add_argument("--prior", choices=["learned", "grid"], default="learned")
    ap.add_argument("--mode", choices=["auto", "eager", "graph"], default="auto",
                    help="eager = stream-ordered launches; graph = HIP-graph replay of the sync-free step (N > 1: five sub-graphs with "
                         "the bucketed all-reduce between them); auto (default) times 3 steps of each after a short warm-up and keeps "
                         "the faster one.  Both overlap the gradient exchange with the backbone's backward")
    ap.add_argument("--no-graph", action="store_true", help="same as --mode eager")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=list(PRECISIONS), default="bf16x3",
                    help="matrix-core arithmetic of the GEMM kernels: split-bf16 x3 (default, ~5e-6 rel) or fp32 MFMA (exact products)")
    ap.add_argument("--no-alt", action="store_true", help="skip the short run in the other precision mode")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra shapes (grid-576 queries, 384x576 image)")
    ap.add_argument("--no-inference", action="store_true", help="skip the inference leg (forward + counting rule, graph replay)")
    ap.add_argument("--no-real-data", action="store_true", help="skip the graph-cache leg (variable image sizes / target counts)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="graph mode: run the frozen stage (stem + layer1) of a batch inside its own step instead of beside the previous step's "
                         "matcher / backward (engine.Trainer: frozen-stage prefetch)")
    a = ap.parse_args(argv)

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        # (ranks started by an external launcher: the HIP runtime has not initialised yet -- nothing above touches the device)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the Counting-DETR HIP path has no CPU fallback")
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world} (launch N ranks, or run `python bench.py --gpus N` and "
                         "let it launch them)")
    # CDETR_BENCH_SHARE_GPU=1: rehearsal of the N>1 control flow on a ONE-GPU box (ranks share the device, gloo carries the
    # collectives through the host).  Never a measurement: the line it prints is tagged "rehearsal".
    share = os.environ.get("CDETR_BENCH_SHARE_GPU", "0") == "1"
    if share:
        local_rank %= torch.cuda.device_count()
    elif world > torch.cuda.device_count():
        raise SystemExit(f"bench.py: {world} ranks but {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if share else "nccl", init_method="env://", world_size=world, rank=rank)   # RCCL over xGMI

    from counting_detr_amd import ops
    H, W = a.size
    Ts = (37, 120)
    if a.no_prefetch:
        os.environ["CDETR_FROZEN_PREFETCH"] = "0"
    trainer = build_trainer(dev, a.queries, a.prior, a.precision)
    Q = a.queries if a.prior == "learned" else int(round(a.queries ** 0.5)) ** 2
    images, rects, targets = synthetic_batch(a.batch, H, W, Ts, seed=1000 * rank, device=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    eager_step = lambda: trainer.train_step(images, rects, targets)      # noqa: E731
    mode = "eager" if a.no_graph else a.mode
    probe = None
    if mode == "auto":      # GPU-bound either way: pick by measurement (the graph saves CPU launch work, the stream path has no per-node overhead)
        def timed(fn, n=3):
            for _ in range(2):
                fn()
            barrier()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            barrier()
            return (time.perf_counter() - t) / n
        # the capture comes FIRST: the trainer probes its side streams at the first capture, and the streams the stream-ordered step creates
        # (weight-gradient side stream, shortcut-branch forks) would otherwise take hardware-queue slots ahead of them -- with two hardware queues
        # the replay then ran 0.6 ms slower (9.29 against 8.69 ms, two same-lease pairs: profiles/r5_ab_hw_queues.txt)
        ok = 1.0
        try:
            trainer.capture(images, rects, targets, warmup=1)
        except Exception as ex:      # a capture problem must not cost the run: the stream-ordered step is always available
            print(f"[bench] graph mode unavailable ({type(ex).__name__}: {ex}); using the eager step", file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            ok = 0.0
        if world > 1:                # every rank times the replay or none does (the replay contains collectives)
            okt = torch.tensor([ok], dtype=torch.float64, device=dev)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            ok = float(okt[0])
        tg = timed(trainer.replay) if ok > 0 else float("inf")
        te = timed(eager_step)
        tt = torch.tensor([te, tg], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)      # every rank takes the same decision
        te, tg = float(tt[0]), float(tt[1])
        mode = "eager" if te <= tg else "graph"
        probe = {"eager_ms": te * 1e3, "graph_ms": tg * 1e3 if tg != float("inf") else None}
    elif mode == "graph":
        trainer.capture(images, rects, targets, warmup=1)
    a.no_graph = (mode == "eager")
    # graph replay of a fixed batch, one step of look-ahead: every step ALSO computes the frozen stage (stem + layer1: no trainable
    # parameter, no gradient) for the step that follows, beside its own Hungarian solve / backward -- what main.py's loop does with the
    # next batch of the loader.  Work per step is unchanged (one frozen stage + one trainable step); --no-prefetch runs it in line.
    pipelined = mode == "graph" and trainer._entry is not None and trainer._entry.get("fs") is not None
    graph_step = (lambda: trainer.replay(pipelined=True)) if pipelined else trainer.replay
    step = eager_step if mode == "eager" else graph_step
    exchange_ab = None
    if world > 1 and mode == "graph" and os.environ.get("CDETR_BENCH_EXCHANGE_AB", "1") != "0":
        exchange_ab = exchange_variants(trainer, graph_step, barrier, dev, share, rank)
    for _ in range(a.warmup):
        out = step()
    if world > 1:
        trainer.exchange.probe = []                        # (compute, comm) event pair per timed step -> exposed all-reduce time
    dt, per_step, out = timed_steps(step, a.steps, barrier)
    exposed = trainer.exchange.exposed_ms() if world > 1 else None
    trainer.exchange.probe = None
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    per_rank_ms = [float(t.item()) / a.steps * 1e3]
    if world > 1:
        mine = torch.tensor([dt / a.steps * 1e3], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_ms = [float(x[0]) for x in allr]
    dt = float(t.item())
    loss = float(out["loss"])
    ms_per_step = dt / a.steps * 1e3
    value = a.batch * world * a.steps / dt

    # ---- roofline leg: same fwd+bwd, eager, every matrix-core launch bracketed by HIP events on its own stream
    import counting_detr_amd.backbone as bb
    hook = bb._BACKWARD_HOOK
    bb.set_backward_hook(None)
    nb = float(sum(len(t_["boxes"]) for t_ in targets))
    from counting_detr_amd.misc import nested_tensor_from_tensor_list
    im, mk = nested_tensor_from_tensor_list(images).decompose()
    reps = 2
    prof = []
    with ops.arithmetic(*trainer.arith), ops.scope(PROFILE=prof):      # (ops.scope: the switch is set for the block and restored whatever happens inside)
        for _ in range(reps):
            trainer._fwd_bwd(im, mk, rects, targets, nb)
    torch.cuda.synchronize()
    fam = {}
    shapes = {}
    for family, flops, e0, e1, tag, nbytes, issued in prof:
        if tag is not None:
            sh = shapes.setdefault((family,) + tuple(tag), [0.0, 0.0, 0])
            sh[0] += flops; sh[1] += e0.elapsed_time(e1) * 1e-3; sh[2] += 1
        f = fam.setdefault(family, [0.0, 0.0, 0, 0.0, 0.0])
        f[4] += issued
        f[0] += flops
        f[1] += e0.elapsed_time(e1) * 1e-3
        f[2] += 1
        f[3] += nbytes
    # the same leg with the weight gradients submitted at the END of the backward instead of beside the data-gradient chain: the family
    # times of kernels that do not share the chip with another stream (the product runs the overlapped form: ops.WGRAD_EVERY)
    unoverlapped = None
    if ops.WGRAD_EVERY or ops.BRANCH_BESIDE:
        prof2 = []
        with ops.arithmetic(*trainer.arith), ops.scope(PROFILE=prof2, WGRAD_EVERY=0, BRANCH_BESIDE=0):
            for _ in range(reps):
                trainer._fwd_bwd(im, mk, rects, targets, nb)
        torch.cuda.synchronize()
        f2 = {}
        for family, flops, e0, e1, tag, nbytes, issued in prof2:
            f = f2.setdefault(family, [0.0, 0.0, 0])
            f[0] += flops; f[1] += e0.elapsed_time(e1) * 1e-3; f[2] += 1
        dense_ = PEAK_FP32_MFMA_TFLOPS if a.precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
        unoverlapped = {k: {"tflops": v[0] / v[1] / 1e12, "frac": v[0] / v[1] / 1e12 / dense_, "ms_per_step": v[1] / reps * 1e3,
                            "avg_launch_us": v[1] / max(v[2], 1) * 1e6} for k, v in f2.items() if v[1] > 0}
        unoverlapped["note"] = ("the same launches with ops.WGRAD_EVERY = 0 and ops.BRANCH_BESIDE = 0 (weight gradients after the data-gradient chain, shortcut "
                                "convolutions in line: nothing shares the chip): per-kernel figures without the stretch of running beside another stream -- "
                                "the step is 0.2-0.3 ms slower that way.  roofline.achieved / frac / families are the launches AS THEY RUN in the product "
                                "(two streams share the chip during the backbone's backward and at its three shortcut convolutions: every kernel there "
                                "takes longer, their sum takes less)")
    if os.environ.get("CDETR_BENCH_SHAPES") and rank == 0:
        rows = sorted(shapes.items(), key=lambda kv: -kv[1][1])
        with open(os.environ["CDETR_BENCH_SHAPES"], "w") as f:
            f.write("family,M,N,K,taps,layout,batch,calls_per_step,us_per_call,ms_per_step,tflops\n")
            for k, v in rows:
                f.write(",".join(str(x) for x in k) + f",{v[2] // reps},{v[1] / v[2] * 1e6:.1f},{v[1] / reps * 1e3:.3f},{v[0] / v[1] / 1e12:.1f}\n")
    bb.set_backward_hook(hook)
    # HBM traffic per launch: PMC counters need their own rocprofv3 passes (FETCH_SIZE and WRITE_SIZE cannot share one), so the
    # figures are read from the committed summary of those passes (tools/run_meas.sh -> profiles/r2_traffic.json)
    tj = None
    if os.path.exists(TRAFFIC_JSON) and (H, W, a.queries, a.batch, a.precision, a.prior) == (800, 800, 300, 2, "bf16x3", "learned"):
        with open(TRAFFIC_JSON) as f:
            tj = json.load(f)
    kern = {}
    for k, v in fam.items():
        if v[1] <= 0:
            continue
        tps = (tj or {}).get("families_bytes_per_step", {}).get(k)         # PMC bytes of the family per step (profiles/r2_traffic.json)
        kern[k] = {"tflops": v[0] / v[1] / 1e12, "mfma_issued_tflops": v[4] / v[1] / 1e12, "mfma_per_product": v[4] / max(v[0], 1.0),
                   "frac": v[0] / v[1] / 1e12 / (PEAK_FP32_MFMA_TFLOPS if a.precision == "fp32" else PEAK_BF16_MFMA_TFLOPS),
                   "mfma_issue_frac": v[4] / v[1] / 1e12 / (PEAK_FP32_MFMA_TFLOPS if a.precision == "fp32" else PEAK_BF16_MFMA_TFLOPS),
                   "traffic_GBps": (tps / (v[1] / reps) / 1e9) if tps else None,
                   "ms_per_step": v[1] / reps * 1e3, "launches_per_step": v[2] // reps,
                   "gflop_per_step": v[0] / reps / 1e9, "algorithmic_bytes_per_launch": v[3] / max(v[2], 1),
                   "algorithmic_bytes_per_step": v[3] / reps, "traffic_bytes_per_step": tps,
                   "traffic_bytes_per_launch": (tps / max(v[2] // reps, 1)) if tps else None,     # per host-side launch (a grouped launch = one)
                   "traffic_over_algorithmic": (tps / (v[3] / reps)) if (tps and v[3]) else None}
    ig = fam.get("igemm", [0.0, 1.0, 1, 0.0, 1.0])
    achieved = ig[0] / ig[1] / 1e12
    # peak for ALGORITHMIC FLOPs of this family = dense bf16 MFMA peak / (bf16 MFMAs issued per algorithmic product, FLOP-weighted
    # over the family's launches: 3 in the split-bf16 forward, 1 in the plain-bf16 backward) -- so frac = issued MFMA rate / 2500 TF
    per_product = ig[4] / max(ig[0], 1.0)
    gflop_img = step_gflop_per_image(H, W, Q)
    dense = PEAK_FP32_MFMA_TFLOPS if a.precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
    total_traffic = (tj or {}).get("total_bytes_per_step")
    roofline = {"bound": "mfma", "achieved": achieved, "peak": dense, "unit": "TFLOP/s",
                "bound_note": "priced against the matrix pipe as the contract asks (dense contraction).  What the launches actually run into at these shapes "
                              "(M = 5000-20000 rows: 1-2.5 tile workgroups per CU) is not the pipe: the long reductions (3x3, K >= 1024) move their operand "
                              "lines L2 -> LDS at 50-75 GB/s per CU (k-steps of 16 / 24 / 32 KB in 0.38 / 0.54 / 0.73 us whatever the loader, ring depth or "
                              "arithmetic), the short ones (K <= 512 into N >= 1024) stream 53-120 MB of epilogue operands at 2.3-2.7 TB/s -- HISTORY.md (round 4), "
                              "profiles/r4_dl_sweep_regstage*.txt, r4_stride_pad.txt, r4_ab_epilogue_touch.txt.  Round 5: a third fewer operand bytes per "
                              "k-step (3x3 halo resident in LDS), another operand format (interleaved groups), 2-4x the work per barrier and 5 instead of 3 "
                              "resident workgroups per CU each left these launches where they were: at about one wave per SIMD a workgroup is a latency "
                              "chain, not a byte stream -- DESIGN.md section 0, profiles/r5_ab_halo_3x3.txt, r5_ab_groups.txt, r5_ab_wgrad.txt",
                "frac": achieved / dense,
                "frac_note": "ALGORITHMIC FLOPs (2*M*N*K*taps per launch) of the dominant kernel family / HIP-event time of its launches / the "
                             "dense MFMA peak of the arithmetic's input type (2500 TF bf16; 157.3 TF for --precision fp32).  The split-bf16 forward "
                             "issues 3 bf16 MFMAs per algorithmic product, the plain-bf16 backward 1: mfma_issue_frac is the issued-MFMA rate "
                             "over the same peak (round 2 reported THAT number as frac)",
                "mfma_per_product": per_product, "mfma_issue_frac": achieved * per_product / dense,
                "traffic": kern.get("igemm", {}).get("traffic_bytes_per_launch"), "traffic_unit": "bytes/launch",
                "traffic_source": (tj or {}).get("source"),
                "algorithmic_bytes_per_launch": ig[3] / max(ig[2], 1),
                "hbm": ({"bytes_per_step": total_traffic, "achieved_TBps": total_traffic / (ms_per_step * 1e-3) / 1e12,
                         "frac_of_8TBps": total_traffic / (ms_per_step * 1e-3) / 8e12,
                         "note": "all kernels of one captured step, FETCH_SIZE (x2, gfx950) + WRITE_SIZE from separate --pmc passes"}
                        if total_traffic else None),
                "kernel": "the tile GEMM kernels (igemm_fast_kernel / igemm_dl_kernel: conv fwd / dgrad / linears of > 48 output tiles) -- algorithmic "
                          "FLOPs 2*M*N*K*taps per launch; the few-row GEMMs (decoder, positional MLPs, heads: launch-latency class, igemm_direct_kernel) "
                          "are the separate family igemm_fewrow",
                "avg_launch_us": ig[1] / max(ig[2], 1) * 1e6, "families": kern, "unoverlapped": unoverlapped,
                "whole_step_tflops": gflop_img * a.batch / ms_per_step, "whole_step_frac": gflop_img * a.batch / ms_per_step / dense,
                "step_gflop_per_image": gflop_img,
                "pmc": (tj or {}).get("pmc")}

    res = {"metric": "images/sec FSCD-147 2nd-stage train step", "value": value, "unit": "images/s", "n_gpus": world,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": dtype_string(a.precision),
           "data": "synthetic" if not share else "synthetic (REHEARSAL: ranks share one GPU, gloo -- not a measurement)",
           "config": {"workload": f"FSCD-147 2nd-stage train step (ResNet-50-DC5 + RCDA enc6/dec6, Q={Q} {a.prior}, "
                                  f"{H}x{W}, T={list(Ts)}), fwd+matcher+loss+bwd+clip+AdamW",
                      "images_per_gpu": a.batch, "global_batch": a.batch * world, "parallelism": f"dp{world}",
                      "graph": not a.no_graph, "mode_probe_ms": probe, "precision": a.precision,
                      "precision_backward": ({1: "bf16x3", 2: "bf16x2", 3: "bf16"}[ops.PRECISION_BWD] if a.precision != "fp32" else "fp32"),
                      "final_loss": loss,
                      "frozen_stage_prefetch": bool(pipelined and mode != "eager")},
           "step_ms": percentiles(per_step),
           # stream -> hardware-queue verdict of the probes (engine.Trainer._side_streams): how many candidate streams run beside the main one,
           # whether the weight-gradient / exchange streams overlap it and each other, and the fail-safe taken when none does
           "streams": getattr(trainer, "side_stream_probe", None),
           "per_rank_ms_per_step": per_rank_ms,
           "scaling_note": ("measured on %d ranks" % world) if world > 1 else
                           "single-GPU line; no N > 1 scaling curve has been measured for this repo yet (one-GPU leases only: the driver's "
                           "SCALE run is the first RCCL execution of the N > 1 path)",
           "roofline": roofline}
    if pipelined and mode != "eager":
        # the same captured step with its frozen stage in line (no look-ahead): short run on the same box, for transparency
        for _ in range(3):
            trainer.replay()
        dtn, pern, _ = timed_steps(trainer.replay, 10, barrier)
        res["frozen_stage_prefetch"] = {
            "what": "graph replay with one batch of look-ahead: a step also runs the frozen stage (stem + max-pool + layer1: no trainable parameter, "
                    "the images need no gradient -- A2/models/backbone.py:93-95) of the batch that FOLLOWS, as its own linear graph on its own "
                    "(probed-concurrent) stream, released by a signal kernel at the head of the step's [Hungarian solve + criterion + backward] "
                    "graph: it runs beside the solve (one wavefront per image, chip otherwise idle).  Work per step is unchanged: one frozen "
                    "stage + one trainable step; `value` is timed this way (main.py's loop does the same with the loader's next batch); "
                    "in_line_*: the same captured step with the stage run inside it (no look-ahead)",
            "graph_layout": trainer._entry.get("layout"), "side_streams": getattr(trainer, "side_stream_probe", None),
            "in_line_ms_per_step": dtn / 10 * 1e3, "in_line_step_ms": percentiles(pern), "prefetched_ms_per_step": ms_per_step,
            "hits": trainer.prefetch_stats["hits"], "in_line_runs": trainer.prefetch_stats["inline"]}
    if world > 1:
        ex = torch.tensor([sum(exposed) / max(len(exposed), 1) if exposed else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(ex, op=dist.ReduceOp.MAX)
        res["exchange_variants"] = exchange_ab
        res["allreduce_exposed_ms"] = {"mean_max_over_ranks": float(ex[0]), "rank0": percentiles(exposed or []),
                                       "bytes_per_step": int(trainer.flat_g.numel() * 4),
                                       "buckets_bytes": [int((trainer.seg_bounds[i + 1] - trainer.seg_bounds[i]) * 4) for i in range(4)],
                                       "backend": "gloo (rehearsal)" if share else "nccl (RCCL over xGMI)",
                                       "note": "time the compute stream waits for the last gradient bucket before clip + AdamW "
                                               "(HIP events on the compute and exchange streams); 0 = fully hidden behind backward"}
    if world == 1 and not a.no_alt:
        # the same step in the other arithmetic mode (short run, same launch mode), for transparency
        alt = "fp32" if a.precision != "fp32" else "bf16x3"
        own, trainer.arith = trainer.arith, (PRECISIONS[alt], trainer.arith[1])      # (the trainer computes in ITS arithmetic, not the module default)
        if a.no_graph:
            alt_step = eager_step
        else:
            trainer.capture(images, rects, targets, warmup=1)
            alt_step = trainer.replay
        for _ in range(2):
            alt_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            alt_step()
        torch.cuda.synchronize()
        dta = (time.perf_counter() - t1) / 5
        res["alt_precision"] = {"precision": alt, "value": a.batch / dta, "unit": "images/s", "ms_per_step": dta * 1e3, "steps": 5}
        trainer.arith = own
    if world == 1 and not a.no_extra:
        del trainer
        torch.cuda.empty_cache()
        res["extra_shapes"] = [
            # the shipped training script: --spatial_prior grid --num_query_position 600 -> 24 x 24 = 576 anchors (A2/scripts/var_wh_laplace_600.sh)
            extra_shape(dev, 800, 800, 600, "grid", Ts, a.batch, a.precision),
            # a typical FSC-147 image after the resize rule (A2/data/fsc147.py:75-77)
            extra_shape(dev, 384, 576, 300, "learned", Ts, a.batch, a.precision),
            # crowded images (FSC-147 holds up to 3731 objects per image, A2/data/fsc147.py:80-84): target-capacity classes 2048 and 3800, where the
            # assignment is one LDS-resident workgroup per image (A2/models/matcher.py:229-247 calls scipy there: 17 ms at 900 x 3000)
            extra_shape(dev, 384, 576, 300, "learned", (37, 2100), a.batch, a.precision),
            extra_shape(dev, 384, 576, 300, "learned", (3000, 3731), a.batch, a.precision),
            # BASELINE configs[3] (FSCD-LVIS 2nd stage) at the largest image its reader feeds (L2/data/fscd_lvis.py:66-91: floor-32 of the image,
            # longest side up to 1333): feature map 50 x 84 -- wider than one 64-key tile (round 6: rcda_fwd2_kernel<4, 6>, dV key-column chunks)
            extra_shape(dev, 800, 1333, 300, "learned", Ts, a.batch, a.precision)]
    if world == 1 and not a.no_inference:
        res["inference"] = inference_leg(dev, [(800, 800), (384, 576), (800, 800, 8), (384, 576, 16)], a.batch, a.precision)
        res["stage1_pseudo_labels"] = stage1_leg(dev, a.precision)
    if world == 1 and not a.no_real_data:
        res["real_data_loop"] = real_data_leg(dev, a.precision, a.batch)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(a.batch, H, W, Ts)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return res


if __name__ == "__main__":
    main()
