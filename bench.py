"""bench.py -- images/sec of one FSCD-147 2nd-stage Counting-DETR training step on N MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = forward + device Hungarian matcher + SetCriterion + backward + clip_grad_norm(0.1) + AdamW on a synthetic
batch of 2 images 800x800 per GPU (Q=300 learned anchors, T=(37,120) targets -- BASELINE.json configs[1]),
inputs resident in HBM, random-init (name-seeded) weights.  Weak scaling: per-GPU batch fixed, gradients averaged over
ranks by RCCL.  The step is replayed from a HIP graph (N=1: one graph; N>1: graph / flat all-reduce / graph).
Prints ONE JSON line on rank 0 with the extra objects `roofline` (fp32-MFMA family of implicit-GEMM kernels, timed with
HIP events on the launch stream in an instrumented eager pass of the same step) and, at N=1, `cpu_baseline` (the oracle's
CPU restatement of the same step on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0       # dense bf16 MFMA; the split-bf16 mode spends 3 bf16 MFMAs per algorithmic product
PRECISIONS = {"bf16x3": 1, "fp32": 0}
STEP_GFLOP_PER_IMAGE = 616.0         # 800x800, Q=300, reduced form (mean-before-project keys)


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


def cpu_baseline(B, H, W, Ts, steps=2, warmup=1):
    """The oracle (CPU restatement of the reference step, parity-pinned against the real reference) on the host cores."""
    from oracle.step import OracleTrainer, synthetic_batch as sb
    cores = os.cpu_count() or 1
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or cores
    except Exception:
        pass
    torch.set_num_threads(cores)
    tr = OracleTrainer(num_position=300)
    images, rects, targets = sb(B=B, H=H, W=W, Ts=Ts)
    for _ in range(warmup):
        tr.step(images, rects, targets)
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(images, rects, targets)
    dt = (time.perf_counter() - t0) / steps
    return {"value": B / dt, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} timed steps (+{warmup} warm-up) of the same B={B} {H}x{W} Q=300 T={list(Ts)} step, oracle fp32 on CPU",
            "ms_per_step": dt * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, nargs=2, default=[800, 800])
    ap.add_argument("--batch", type=int, default=2, help="images per GPU")
    ap.add_argument("--queries", type=int, default=300)
    ap.add_argument("--mode", choices=["auto", "eager", "graph"], default="auto",
                    help="eager = stream-ordered launches (bucketed all-reduce overlapped with backward); graph = HIP-graph replay of the "
                         "sync-free step; auto (default) times 3 steps of each after a short warm-up and keeps the faster one")
    ap.add_argument("--no-graph", action="store_true", help="same as --mode eager")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=list(PRECISIONS), default="bf16x3",
                    help="matrix-core arithmetic of the GEMM kernels: split-bf16 x3 (default, ~5e-6 rel) or fp32 MFMA (exact products)")
    ap.add_argument("--no-alt", action="store_true", help="skip the short run in the other precision mode")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the Counting-DETR HIP path has no CPU fallback")
    # CDETR_BENCH_SHARE_GPU=1: rehearsal of the N>1 control flow on a ONE-GPU box (ranks share the device, gloo carries the
    # collectives through the host).  Never a measurement: the line it prints is tagged "rehearsal".
    share = os.environ.get("CDETR_BENCH_SHARE_GPU", "0") == "1"
    if share:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if share else "nccl", init_method="env://", world_size=world, rank=rank)   # RCCL over xGMI
    assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    import counting_detr_amd
    from counting_detr_amd import ops
    from counting_detr_amd.args import default_args
    from counting_detr_amd.engine import Trainer
    from counting_detr_amd.init import seeded_init_

    ops.PRECISION = PRECISIONS[a.precision]
    H, W = a.size
    Ts = (37, 120)
    args = default_args(device=str(dev), num_query_position=a.queries)
    model, crit, _ = counting_detr_amd.build_model(args)
    seeded_init_(model)          # deterministic name-seeded random weights (no checkpoints in this environment)
    model.to(dev).train()
    crit.train()
    trainer = Trainer(model, crit, args, device=dev)
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
        te = timed(eager_step)
        ok = 1.0
        try:
            trainer.capture(images, rects, targets, warmup=1)
        except Exception as ex:      # a capture problem must not cost the run: the stream-ordered step is always available
            print(f"[bench] graph mode unavailable ({type(ex).__name__}: {ex}); using the eager step", file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            ok = 0.0
        if world > 1:                # every rank times the replay or none does (the replay contains a collective)
            okt = torch.tensor([ok], dtype=torch.float64, device=dev)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            ok = float(okt[0])
        tg = timed(trainer.replay) if ok > 0 else float("inf")
        tt = torch.tensor([te, tg], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)      # every rank takes the same decision
        te, tg = float(tt[0]), float(tt[1])
        mode = "eager" if te <= tg else "graph"
        probe = {"eager_ms": te * 1e3, "graph_ms": tg * 1e3 if tg != float("inf") else None}
    elif mode == "graph":
        trainer.capture(images, rects, targets, warmup=1)
    a.no_graph = (mode == "eager")
    step = eager_step if mode == "eager" else trainer.replay
    for _ in range(a.warmup):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    loss = float(out["loss"])
    ms_per_step = dt / a.steps * 1e3
    value = a.batch * world * a.steps / dt

    # ---- roofline leg: same fwd+bwd, eager, every matrix-core launch bracketed by HIP events on its own stream
    import counting_detr_amd.backbone as bb
    hook = bb._BACKWARD_HOOK
    bb.set_backward_hook(None)
    ops.PROFILE = []
    nb = float(sum(len(t_["boxes"]) for t_ in targets))
    from counting_detr_amd.misc import nested_tensor_from_tensor_list
    im, mk = nested_tensor_from_tensor_list(images).decompose()
    reps = 2
    for _ in range(reps):
        trainer._fwd_bwd(im, mk, rects, targets, nb)
    torch.cuda.synchronize()
    fam = {}
    shapes = {}
    ig_alg_bytes = 0.0
    for family, flops, e0, e1, tag in ops.PROFILE:
        if family == "igemm" and tag is not None and tag[0] == "group":
            ig_alg_bytes += tag[1]           # grouped launch: the sum over its problems (ops.gemm_queue)
        elif family == "igemm" and tag is not None:
            M_, N_, K_, taps_ = tag[0], tag[1], tag[2], tag[3]
            # compulsory fp32 bytes of one launch: input rows once (a strided/dilated conv reads <= M*K of them), weights, output
            ig_alg_bytes += 4.0 * (M_ * K_ + N_ * K_ * taps_ + M_ * N_) * max(tag[5], 1)
        if tag is not None:
            sh = shapes.setdefault((family,) + tuple(tag), [0.0, 0.0, 0])
            sh[0] += flops; sh[1] += e0.elapsed_time(e1) * 1e-3; sh[2] += 1
        f = fam.setdefault(family, [0.0, 0.0, 0])
        f[0] += flops
        f[1] += e0.elapsed_time(e1) * 1e-3
        f[2] += 1
    ops.PROFILE = None
    if os.environ.get("CDETR_BENCH_SHAPES") and rank == 0:
        rows = sorted(shapes.items(), key=lambda kv: -kv[1][1])
        with open(os.environ["CDETR_BENCH_SHAPES"], "w") as f:
            f.write("family,M,N,K,taps,layout,batch,calls_per_step,us_per_call,ms_per_step,tflops\n")
            for k, v in rows:
                f.write(",".join(str(x) for x in k) + f",{v[2] // reps},{v[1] / v[2] * 1e6:.1f},{v[1] / reps * 1e3:.3f},{v[0] / v[1] / 1e12:.1f}\n")
    bb.set_backward_hook(hook)
    kern = {k: {"tflops": v[0] / v[1] / 1e12, "ms_per_step": v[1] / reps * 1e3, "launches_per_step": v[2] // reps,
                "gflop_per_step": v[0] / reps / 1e9} for k, v in fam.items() if v[1] > 0}
    ig = fam.get("igemm", [0.0, 1.0, 1])
    achieved = ig[0] / ig[1] / 1e12
    peak = PEAK_FP32_MFMA_TFLOPS if a.precision == "fp32" else PEAK_BF16_MFMA_TFLOPS / 3.0
    # HBM traffic per launch of the same family: PMC counters need their own rocprofv3 passes (FETCH_SIZE and WRITE_SIZE
    # cannot share one), so the figure is read from the committed summary of those passes (tools/run_meas.sh).
    traffic = traffic_src = None
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_traffic.json")
    if os.path.exists(tpath) and (H, W, a.queries, a.batch, a.precision) == (800, 800, 300, 2, "bf16x3"):
        with open(tpath) as f:
            tj = json.load(f)
        traffic, traffic_src = tj["bytes_per_launch"], tj["source"]
    roofline = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak, "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": ig_alg_bytes / max(ig[2], 1),
                "peak_note": ("fp32 MFMA v_mfma_f32_32x32x2_f32" if a.precision == "fp32" else
                              "2500 TF dense bf16 MFMA / 3 MFMAs per algorithmic product (hi*hi + hi*lo + lo*hi)"),
                "kernel": "igemm_fast_kernel (conv fwd / dgrad / linear) -- algorithmic FLOPs 2*M*N*K*taps per launch",
                "avg_launch_us": ig[1] / max(ig[2], 1) * 1e6, "families": kern,
                "whole_step_tflops": STEP_GFLOP_PER_IMAGE * a.batch / ms_per_step if (H, W, a.queries) == (800, 800, 300) else None}

    res = {"metric": "images/sec FSCD-147 2nd-stage train step", "value": value, "unit": "images/s", "n_gpus": world,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": ("fp32" if a.precision == "fp32" else "bf16x3 (fp32 operands split hi+lo, 3 bf16 MFMAs per product, fp32 accumulate; fp32 storage)"),
           "data": "synthetic" if not share else "synthetic (REHEARSAL: ranks share one GPU, gloo -- not a measurement)",
           "config": {"workload": f"FSCD-147 2nd-stage train step (ResNet-50-DC5 + RCDA enc6/dec6, Q={a.queries} learned, "
                                  f"{H}x{W}, T={list(Ts)}), fwd+matcher+loss+bwd+clip+AdamW",
                      "images_per_gpu": a.batch, "global_batch": a.batch * world, "parallelism": f"dp{world}",
                      "graph": not a.no_graph, "mode_probe_ms": probe, "precision": a.precision, "final_loss": loss},
           "roofline": roofline}
    if world == 1 and not a.no_alt:
        # the same step in the other arithmetic mode (short run, same launch mode), for transparency
        alt = "fp32" if a.precision != "fp32" else "bf16x3"
        ops.PRECISION = PRECISIONS[alt]
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
        ops.PRECISION = PRECISIONS[a.precision]
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(a.batch, H, W, Ts)
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
