"""Host helpers mirrored from A2/util/misc.py: NestedTensor (:313-336), nested_tensor_from_tensor_list (:291-310),
distributed helpers (:339-433), reduce_dict (:133-157), accuracy (:436-452)."""
import os

import torch
import torch.distributed as dist


class NestedTensor(object):
    def __init__(self, tensors, mask):
        self.tensors = tensors
        self.mask = mask

    def to(self, device, non_blocking=False):
        m = self.mask.to(device, non_blocking=non_blocking) if self.mask is not None else None
        return NestedTensor(self.tensors.to(device, non_blocking=non_blocking), m)

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self):
        return str(self.tensors)


_NO_PADDING = {}


def nested_tensor_from_tensor_list(tensor_list):
    """Zero-pad a list of [3,h,w] images (or the rows of a [B,3,H,W] tensor) to a common size + bool padding mask."""
    if isinstance(tensor_list, torch.Tensor):
        b, c, h, w = tensor_list.shape
        key = (b, h, w, str(tensor_list.device))
        m = _NO_PADDING.get(key)           # a batch given as ONE tensor has no padding: the all-false mask is built once per shape
        if m is None:
            if len(_NO_PADDING) > 16:
                _NO_PADDING.clear()
            m = _NO_PADDING[key] = torch.zeros((b, h, w), dtype=torch.bool, device=tensor_list.device)
        return NestedTensor(tensor_list, m)
    if tensor_list[0].ndim != 3:
        raise ValueError("not supported")
    hh = max(img.shape[1] for img in tensor_list)
    ww = max(img.shape[2] for img in tensor_list)
    b, c = len(tensor_list), tensor_list[0].shape[0]
    dtype, device = tensor_list[0].dtype, tensor_list[0].device
    tensor = torch.zeros((b, c, hh, ww), dtype=dtype, device=device)
    mask = torch.ones((b, hh, ww), dtype=torch.bool, device=device)
    for img, pad_img, m in zip(tensor_list, tensor, mask):
        pad_img[: img.shape[0], : img.shape[1], : img.shape[2]].copy_(img)
        m[: img.shape[1], : img.shape[2]] = False
    return NestedTensor(tensor, mask)


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


def init_distributed_mode(args):
    """env:// rendezvous, one process per GPU; backend "nccl" is RCCL on ROCm (A2/util/misc.py:396-433)."""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank = int(os.environ["RANK"])
        args.world_size = int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ.get("LOCAL_RANK", 0))
    else:
        args.distributed = False
        return
    args.distributed = True
    backend = getattr(args, "dist_backend", "nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(args.gpu)
    dist.init_process_group(backend=backend, init_method=getattr(args, "dist_url", "env://"),
                            world_size=args.world_size, rank=args.rank)
    dist.barrier()


def reduce_dict(input_dict, average=True):
    """One all-reduce over the stacked scalar losses (A2/util/misc.py:133-157)."""
    world_size = get_world_size()
    if world_size < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k].detach().float() for k in names], dim=0)
        dist.all_reduce(values)
        if average:
            values /= world_size
        return {k: v for k, v in zip(names, values)}


@torch.no_grad()
def accuracy(output, target, topk=(1,)):
    if target.numel() == 0:
        return [torch.zeros([], device=output.device)]
    maxk = max(topk)
    _, pred = output.topk(maxk, 1, True, True)
    pred = pred.t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    return [correct[:k].reshape(-1).float().sum(0).mul_(100.0 / target.size(0)) for k in topk]
