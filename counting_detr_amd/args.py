"""The reference's CLI (A2/main.py:17-135): same flag names and defaults, so scripts written for the reference's
main.py keep working.  `default_args(**overrides)` builds the namespace programmatically."""
import argparse


def get_args_parser():
    p = argparse.ArgumentParser("Counting-DETR 2nd stage (MI355X)", add_help=True)
    p.add_argument("-dp", "--data_path", type=str, default="./FSC147/")
    p.add_argument("-o", "--output_dir", type=str, default="./outputs/anchor_detr")
    p.add_argument("-ts", "--test_split", type=str, default="val", choices=["val_PartA", "val_PartB", "test_PartA", "test_PartB", "test", "val"])
    p.add_argument("--lr", default=1e-4, type=float)
    p.add_argument("--lr_backbone_names", default=["backbone"], type=str, nargs="+")
    p.add_argument("--lr_backbone", default=1e-5, type=float)
    p.add_argument("--lr_linear_proj_names", default=[], type=str, nargs="+")
    p.add_argument("--lr_linear_proj_mult", default=0.1, type=float)
    p.add_argument("--batch_size", default=1, type=int)
    p.add_argument("--weight_decay", default=1e-4, type=float)
    p.add_argument("--epochs", default=30, type=int)
    p.add_argument("--lr_drop", default=20, type=int)
    p.add_argument("--lr_drop_epochs", default=None, type=int, nargs="+")
    p.add_argument("--clip_max_norm", default=0.1, type=float)
    p.add_argument("--sgd", action="store_true")
    p.add_argument("--frozen_weights", type=str, default=None)
    p.add_argument("--backbone", default="resnet50", type=str)
    p.add_argument("--dilation", default=True, type=lambda s: str(s).lower() not in ("0", "false", "no"))
    p.add_argument("--num_feature_levels", default=1, type=int)
    p.add_argument("--enc_layers", default=6, type=int)
    p.add_argument("--dec_layers", default=6, type=int)
    p.add_argument("--dim_feedforward", default=1024, type=int)
    p.add_argument("--hidden_dim", default=256, type=int)
    p.add_argument("--dropout", default=0.0, type=float)
    p.add_argument("--nheads", default=8, type=int)
    p.add_argument("--num_query_position", default=300, type=int)
    p.add_argument("--num_query_pattern", default=3, type=int)
    p.add_argument("--spatial_prior", default="learned", choices=["learned", "grid", "defined"], type=str)
    p.add_argument("--attention_type", default="RCDA", choices=["RCDA", "nn.MultiheadAttention"], type=str)
    p.add_argument("--masks", action="store_true")
    p.add_argument("--no_aux_loss", dest="aux_loss", action="store_false")
    p.add_argument("--cost_class", default=2, type=float)
    p.add_argument("--cost_bbox", default=5, type=float)
    p.add_argument("--cost_giou", default=2, type=float)
    p.add_argument("--mask_loss_coef", default=1, type=float)
    p.add_argument("--dice_loss_coef", default=1, type=float)
    p.add_argument("--cls_loss_coef", default=2, type=float)
    p.add_argument("--bbox_loss_coef", default=5, type=float)
    p.add_argument("--giou_loss_coef", default=2, type=float)
    p.add_argument("--focal_alpha", default=0.25, type=float)
    p.add_argument("--variance_loss_coef", default=2, type=float)
    p.add_argument("--device", default="cuda")
    p.add_argument("--seed", default=42, type=int)
    p.add_argument("--resume", default="")
    p.add_argument("--auto_resume", default=False, action="store_true")
    p.add_argument("--start_epoch", default=0, type=int)
    p.add_argument("--eval", action="store_true")
    p.add_argument("--num_workers", default=2, type=int)
    p.add_argument("--scale_factor", default=32, type=int, help="val / test images are resized to a multiple of this (A2/infer.py)")
    p.add_argument("--split", default="val", type=str, help="infer.py: val or test")
    p.add_argument("--cache_mode", default=False, action="store_true")
    # additions of this build (the reference hard-codes batch 1 on one GPU)
    p.add_argument("--dataset", default="fsc147", choices=["fsc147", "fscd_lvis"], help="reader used without --synthetic")
    p.add_argument("--images_per_gpu", default=2, type=int, help="local batch of the data-parallel trainer")
    p.add_argument("--synthetic", action="store_true", help="train on seeded synthetic tensors (no dataset needed)")
    p.add_argument("--steps_per_epoch", default=20, type=int, help="synthetic mode only")
    p.add_argument("--synthetic_size", default=[800, 800], type=int, nargs=2, help="synthetic mode only: image H W")
    p.add_argument("--pretrained_backbone", default="", type=str,
                   help="torchvision-layout ResNet-50 state dict (the reference hard-codes pretrained_models/resnet50-0676ba61.pth)")
    p.add_argument("--resume_skip_mismatch", action="store_true",
                   help="--resume: drop checkpoint keys whose shape differs (e.g. an Anchor-DETR COCO class head) instead of raising")
    p.add_argument("--resume_optimizer", action="store_true",
                   help="--resume: ALSO restore AdamW moments, StepLR state and the epoch counter from the checkpoint (continue an "
                        "interrupted run).  Default: model weights only and training starts at --start_epoch, exactly like the "
                        "reference (A2/main.py:195-209), which fine-tunes from a full detector checkpoint this way")
    p.add_argument("--no_resume_optimizer", dest="resume_optimizer", action="store_false", help="(default; kept for scripts of round 2)")
    p.add_argument("--no_graph_cache", dest="graph_cache", action="store_false",
                   help="train with the stream-ordered step instead of cached HIP graphs (one per padded image size / target-capacity class)")
    p.add_argument("--graph_cache_size", default=32, type=int, help="captured steps kept alive (least recently used is dropped)")
    p.add_argument("--graph_layout", default="chain", choices=["chain", "single"],
                   help="captured step: chain = linear graphs + side-stream graphs (host cost 0.3 ms per step); single = round 3's two graphs with "
                        "in-graph branches (8.7 ms of host time per step)")
    p.add_argument("--no_frozen_prefetch", dest="frozen_prefetch", action="store_false",
                   help="run the frozen stem + layer1 of a batch inside its own step instead of beside the previous step's Hungarian solve")
    p.add_argument("--captured_allreduce", action="store_true",
                   help="world_size > 1: the four gradient buckets' all-reduces as captured graphs on the exchange stream instead of host-issued "
                        "RCCL calls (untested on N > 1 GPUs: for the first multi-GPU A/B)")
    p.add_argument("--bwd_precision", default=None, choices=["bf16", "bf16x2", "bf16x3"],
                   help="arithmetic of the backward contractions (default bf16 = 1 MFMA per product, gradient error 4.7e-3 of its norm; "
                        "bf16x3 = the forward's split arithmetic, 1.6e-3; DESIGN section 3).  Same as CDETR_PRECISION_BWD=3/2/1")
    p.add_argument("--exemplar_mode", default="per_image", choices=["per_image", "reference"],
                   help="per_image: image b is conditioned on its own exemplars; reference: rects[0] for the whole batch "
                        "(A2/models/backbone.py:122 -- exact only at batch 1)")
    return p


def default_args(**kw):
    a = get_args_parser().parse_args([])
    a.aux_loss = False            # the shipped scripts all pass --no_aux_loss (A2/scripts/var_wh_laplace_600.sh:7)
    a.num_query_pattern = 1
    for k, v in kw.items():
        setattr(a, k, v)
    return a
