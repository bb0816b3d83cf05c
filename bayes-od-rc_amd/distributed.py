"""Image-sharded multi-GPU execution: one process per GPU, images are independent units
(the reference's loop is batch(1) with no cross-image state: run_inference.py:68,137), so the
only exchange is ONE gather of the final fixed-size detection records per step
(SURVEY.md section 8e).  ``torch.distributed`` is plumbing: backend "nccl" is RCCL over xGMI on
ROCm, "gloo" is used by the CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(num_images, world_size, rank):
    """Contiguous split of [0, num_images) over ranks; the first (num_images % world) ranks get one more."""
    base, extra = divmod(num_images, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


RECORD_EXTRA = 1      # per-detection record = [valid, means 4, covs 16, scores C, counts C]


def record_width(num_classes):
    return RECORD_EXTRA + 4 + 16 + 2 * num_classes


def pack_records(num, scores, means, covs, counts):
    """Padded per-image arrays -> one float32 tensor [B, K, 1+4+16+2C]; slot 0 flags valid rows.
    Works on torch tensors of any device (device-side pack before the RCCL gather)."""
    b, k, _ = scores.shape
    valid = (torch.arange(k, device=scores.device)[None, :] < num.to(scores.device)[:, None]).to(scores.dtype)
    rec = torch.cat([valid[:, :, None], means.reshape(b, k, 4), covs.reshape(b, k, 16), scores, counts], dim=2)
    return rec * valid[:, :, None]


def unpack_records(rec, num_classes):
    """[B,K,W] tensor/array -> list of (scores [k,C], means [k,4], covs [k,4,4], counts [k,C]) per image."""
    rec = rec.detach().cpu().numpy() if isinstance(rec, torch.Tensor) else np.asarray(rec)
    out = []
    c = num_classes
    for r in rec:
        k = int(r[:, 0].sum())
        r = r[:k]
        out.append((r[:, 21:21 + c], r[:, 1:5], r[:, 5:21].reshape(k, 4, 4), r[:, 21 + c:21 + 2 * c]))
    return out


def gather_records(rec, dst=0, group=None):
    """One collective per step: every rank contributes its [B,K,W] block; rank ``dst`` receives
    [world, B, K, W] (others None).  Latency-bound (~14 KB per image), never a ring all-reduce."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return rec[None]
    rank = dist.get_rank(group)
    if rank == dst:
        bufs = [torch.empty_like(rec) for _ in range(world)]
        dist.gather(rec, gather_list=bufs, dst=dst, group=group)
        return torch.stack(bufs)
    dist.gather(rec, gather_list=None, dst=dst, group=group)
    return None


class DeviceArray(object):
    """Zero-copy view of a device buffer owned by the HIP library, for torch.as_tensor()."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}


def torch_views(engine, slot=0):
    """torch tensors aliasing the engine's detection buffers (record slot 0/1) on its GPU."""
    p = engine.device_detection_pointers(slot)
    dev = torch.device("cuda", engine.cfg.device)
    views = {}
    for name, (ptr, shape) in p.items():
        views[name] = torch.as_tensor(DeviceArray(ptr, shape, "<i4" if name == "num" else "<f4"), device=dev)
    return views
