"""main.py -- CLI of the reference's 2nd-stage trainer (A2/main.py:17-258) on the MI355X path.

Same flags, same driver behaviour: build_model -> 3 lr groups / AdamW / StepLR(lr_drop) -> optional --resume (model
weights only, keys filtered like A2/main.py:195-209) -> per epoch train_one_epoch, scheduler step, checkpoint
{"model","optimizer","lr_scheduler","epoch","args"} to <output_dir>/detr_retrain.pth (+ numbered copies), JSON log line.
Differences: any number of images per GPU (--images_per_gpu), data-parallel over the GPUs of a node under torchrun
(RCCL), and --synthetic (seeded tensors; FSC-147 is not available in this environment -- with a dataset, pass a
DataLoader yielding the reference's sample dicts to `train_one_epoch`).  Without --synthetic the FSC-147 reader of
counting_detr_amd/data.py feeds the step (batched collate + pinned-memory prefetch).

  python main.py --synthetic --no_aux_loss --num_query_pattern 1 --epochs 1 -o /tmp/out
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 main.py --synthetic --no_aux_loss ...
"""
import json
import os
import time
from pathlib import Path

if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    # data-parallel run: the step uses main + prefetch + weight-gradient + exchange streams and RCCL adds its own -- more than HIP's default
    # four hardware queues (two streams of one queue never overlap).  Must be in the environment before the HIP runtime initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch

import counting_detr_amd
from counting_detr_amd import checkpoint as ckpt_io
from counting_detr_amd import misc as utils
from counting_detr_amd.args import get_args_parser
from counting_detr_amd.engine import Trainer, count_objects, counting_metrics, train_one_epoch


class SyntheticLoader:
    """Seeded stand-in for FSC147Dataset + DataLoader: yields the step's sample dict (images already on the device)."""

    def __init__(self, args, device, steps, size=(800, 800), targets=(37, 120)):
        self.args, self.device, self.steps, self.size, self.targets = args, device, steps, size, targets

    def __iter__(self):
        B = self.args.images_per_gpu
        H, W = self.size
        rects = torch.tensor([[.10, .10, .20, .20], [.40, .40, .50, .55], [.70, .20, .80, .30]], device=self.device)
        for it in range(self.steps):
            g = torch.Generator().manual_seed(1000003 * utils.get_rank() + it)
            tg = []
            for b in range(B):
                T = self.targets[b % len(self.targets)]
                box = torch.cat([torch.rand(T, 2, generator=g) * 0.8 + 0.1, torch.rand(T, 2, generator=g) * 0.10 + 0.02], 1)
                tg.append({"boxes": box.to(self.device), "labels": torch.zeros(T, dtype=torch.int64, device=self.device)})
            yield {"image": torch.randn(B, 3, H, W, generator=g).to(self.device), "ex_rects": rects[None].repeat(B, 1, 1),
                   "targets": tg}

    def __len__(self):
        return self.steps


def main(args):
    utils.init_distributed_mode(args)
    os.makedirs(args.output_dir, exist_ok=True)
    device = torch.device(args.device if not getattr(args, "distributed", False) else f"cuda:{args.gpu}")
    # every rank builds the SAME initial weights (the init draws from the global RNG); only the data order is rank-specific.
    # Trainer additionally broadcasts rank 0's parameters and buffers (what DistributedDataParallel does at construction).
    torch.manual_seed(args.seed)
    # arithmetic of the backward contractions: handed to the trainer, which owns its arithmetic (ops.arithmetic) -- no module global is flipped
    bwd_precision = {"bf16": 3, "bf16x2": 2, "bf16x3": 1}[args.bwd_precision] if getattr(args, "bwd_precision", None) else None
    model, criterion, _ = counting_detr_amd.build_model(args)
    model.to(device)
    if args.pretrained_backbone:                                        # A2/models/backbone.py:153-155 -> resnet.py:292-297
        n = ckpt_io.load_backbone_pretrained(model, args.pretrained_backbone)
        print(f"backbone: {n} tensors from {args.pretrained_backbone}")

    checkpoint = None
    if args.auto_resume and not args.resume:                            # A1/main.py:217-221: continue from the output directory's last checkpoint
        last = os.path.join(args.output_dir, "detr_retrain.pth")
        if os.path.isfile(last):
            args.resume = last
    if args.resume:                                                     # A2/main.py:195-209
        checkpoint, _, _ = ckpt_io.resume_model(model, args.resume, skip_mismatch=args.resume_skip_mismatch)

    trainer = Trainer(model, criterion, args, device=device, precision_bwd=bwd_precision)           # 3 lr groups + flat AdamW (A2/main.py:157-189); syncs replicas; owns its arithmetic
    if checkpoint is not None and (args.resume_optimizer or args.auto_resume) and checkpoint.get("optimizer"):
        # opt-in (the reference loads weights only and starts at --start_epoch, A2/main.py:195-209): continue an interrupted run
        trainer.load_state_dict(checkpoint["optimizer"], checkpoint.get("lr_scheduler"))
        resumed = int(checkpoint.get("epoch", -1)) + 1
        if resumed > args.start_epoch:
            print(f"resume: optimizer state restored, continuing at epoch {resumed} (checkpoint epoch {resumed - 1})")
            args.start_epoch = resumed
    if args.start_epoch >= args.epochs:
        print(f"nothing to train: start epoch {args.start_epoch} >= --epochs {args.epochs}")
    torch.manual_seed(args.seed + 1 + utils.get_rank())                 # data-side RNG (synthetic batches, augmentation)
    sampler = None
    if args.synthetic:
        loader = SyntheticLoader(args, device, args.steps_per_epoch, size=tuple(args.synthetic_size))
    else:                                                               # A2/main.py:146-147 (+ batching, sharding, prefetch)
        from torch.utils.data import DataLoader, DistributedSampler
        from counting_detr_amd import data
        ds = data.build_dataset(args)
        sampler = DistributedSampler(ds, shuffle=True, seed=args.seed) if getattr(args, "distributed", False) else None
        dl = DataLoader(ds, batch_size=args.images_per_gpu, shuffle=(sampler is None), sampler=sampler, collate_fn=data.collate,
                        num_workers=args.num_workers, drop_last=True, pin_memory=False)
        loader = data.Prefetcher(dl, device)
    print("Start training")
    start = time.time()
    output_dir = Path(args.output_dir)
    for epoch in range(args.start_epoch, args.epochs):
        if sampler is not None:
            sampler.set_epoch(epoch)           # a different shuffle (and rank sharding) every epoch
        stats = train_one_epoch(trainer, loader, epoch, print_freq=10)
        trainer.lr_scheduler_step()
        paths = [output_dir / "detr_retrain.pth"]
        if (epoch + 1) % args.lr_drop == 0 or (epoch + 1) % 10 == 0:
            paths.append(output_dir / f"detr_retrain_{epoch:04}.pth")
        if utils.is_main_process():            # only rank 0 pays for the host copy of the moments
            ckpt = {"model": model.state_dict(), "optimizer": trainer.state_dict(), "lr_scheduler": trainer.lr_scheduler_state_dict(),
                    "epoch": epoch, "args": args}
            for p in paths:
                torch.save(ckpt, p)
        if utils.is_main_process():
            with (output_dir / "detr_retrain.txt").open("a") as f:
                f.write(json.dumps({**{f"train_{k}": v for k, v in stats.items()}, "epoch": epoch}) + "\n")
    print("time: ", time.time() - start)
    if args.eval and not args.synthetic:                                # counting evaluation on the validation split (A2/infer.py)
        if utils.is_main_process():
            import infer as _infer
            from torch.utils.data import DataLoader
            from counting_detr_amd import data
            dl = DataLoader(data.build_test_dataset(args, image_set=args.split), batch_size=1, shuffle=False, collate_fn=data.collate,
                            num_workers=args.num_workers)
            metrics, _ = _infer.infer(model, criterion, dl, device, args.output_dir, split=args.split)
            print("counting metrics ({}): {}".format(args.split, json.dumps(metrics)))
    if args.eval and args.synthetic:                                    # counting rule + MAE on the synthetic shard
        pred, gt = [], []
        for ret in SyntheticLoader(args, device, 2):
            counts, _, _, _ = count_objects(model, ret["image"], ret["ex_rects"])
            pred += [int(c) for c in counts]
            gt += [len(t["boxes"]) for t in ret["targets"]]
        print("counting metrics (synthetic):", counting_metrics(pred, gt))


if __name__ == "__main__":
    a = get_args_parser().parse_args()
    if a.output_dir:
        Path(a.output_dir).mkdir(parents=True, exist_ok=True)
    main(a)
