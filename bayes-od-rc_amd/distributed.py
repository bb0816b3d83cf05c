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


def gather_records(rec, dst=0, group=None, always=False):
    """One collective per step: every rank contributes its [B,K,W] block; rank ``dst`` receives
    [world, B, K, W] (others None).  Latency-bound (~14 KB per image), never a ring all-reduce.
    ``always``: issue the collective even in a one-rank group (exercises the backend on a one-GPU box)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (always and dist.is_initialized()):
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


# ---------------------------------------------------------------------------------------------------
# Second mode: MC-sample sharding for single-frame latency.  Every rank runs the
# backbone/FPN of the SAME frame(s) and n = N/world of the N dropout samples (its handle has
# mc_sample_base = rank*n, so the Philox streams are those of samples rank*n .. rank*n+n-1 of the N-sample
# ensemble); ONE all-gather of the raw head outputs rebuilds the [B,N,A,.] tensors RetinaNetModel.call
# returns on every rank, bit-identical to a single-GPU run, and the (cheap) Bayesian stages run replicated.
# The exchange carries 22 floats per anchor and sample ((C + 4 + 10) * 4 B * A * n per rank; 2.6 MB per
# sample at 512x512) over xGMI -- direct all-gather, no reduction, nothing to re-associate.
# ---------------------------------------------------------------------------------------------------
def sample_shard(total_samples, world_size, rank):
    """(first sample, count) of this rank; the ensemble must split evenly (fixed-size all-gather)."""
    if total_samples % world_size != 0:
        raise ValueError("mc_dropout_samples=%d is not divisible by the %d ranks of the sample-sharded mode"
                         % (total_samples, world_size))
    n = total_samples // world_size
    return rank * n, n


def raw_views(engine, mark_ready=False):
    """torch tensors aliasing the engine's raw head-output buffers: cls [B,N,A,C], box [B,N,A,4], cov [B,N,A,10]."""
    ptrs = engine.device_raw_pointers(mark_ready)
    dev = torch.device("cuda", engine.cfg.device)
    b, n, a = engine.B, engine.N, engine.A
    shapes = {"cls": (b, n, a, engine.Ccls), "box": (b, n, a, 4), "cov": (b, n, a, 10)}
    return {k: torch.as_tensor(DeviceArray(p, shapes[k], "<f4"), device=dev)
            for k, p in zip(("cls", "box", "cov"), ptrs) if p}


def all_gather_samples(local, full, group=None):
    """local [B,n,A,c] of every rank -> full [B,world*n,A,c] on every rank, rank r's samples at r*n.. .
    B == 1 gathers straight into ``full`` (zero copy); B > 1 goes through one staging tensor."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    b, n = local.shape[0], local.shape[1]
    if tuple(full.shape) != (b, world * n) + tuple(local.shape[2:]):
        raise ValueError("full %s does not hold %d x local %s" % (tuple(full.shape), world, tuple(local.shape)))
    if world == 1:
        full.copy_(local)
        return full
    local = local.contiguous()
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # gloo moves device tensors through the host anyway; used by the one-GPU test of this mode (RCCL in production)
        host = torch.empty(tuple(full.shape), dtype=full.dtype)
        all_gather_samples(local.cpu(), host, group)
        full.copy_(host)
        return full
    if b == 1:
        dist.all_gather_into_tensor(full.view((world * n,) + tuple(local.shape[2:])),
                                    local.view((n,) + tuple(local.shape[2:])), group=group)
    else:
        tmp = torch.empty((world * b,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(tmp, local, group=group)           # concatenation along dim 0: [world*B, n, A, c]
        full.view((b, world, n) + tuple(local.shape[2:])).copy_(tmp.view((world,) + tuple(local.shape)).transpose(0, 1))
    return full


class SampleShardedEngine(object):
    """Pair of handles for the sample-sharded mode on this rank's GPU: ``fwd`` computes this rank's n samples,
    ``post`` (no weights: raw buffers only) receives the gathered ensemble and runs posterior / soft-NMS /
    cluster-fuse.  ``make_config_kwargs`` are those of engine.make_config for the FULL ensemble."""

    def __init__(self, image_hw, weights, anchors, mc_samples, device=0, batch=1, group=None, **make_config_kwargs):
        from .engine import Engine, make_config
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        base, n = sample_shard(mc_samples, self.world, self.rank)
        self.fwd = Engine(make_config(image_hw, batch=batch, mc_samples=n, device=device, mc_sample_base=base,
                                      mc_ensemble_size=mc_samples, **make_config_kwargs))
        self.fwd.load_weights(weights)
        self.post = Engine(make_config(image_hw, batch=batch, mc_samples=mc_samples, device=device, **make_config_kwargs))
        self.post.set_anchors(anchors)
        self._local = raw_views(self.fwd)
        self._full = raw_views(self.post, mark_ready=False)

    def infer(self, images, seed=0, first_image_id=0):
        """images: [B,H,W,3] float32 (the same on every rank).  Returns this rank's copy of the detections
        (list over images of (scores, means, covs, counts)); identical on all ranks."""
        self.fwd.forward(images, seed=seed, first_image_id=first_image_id)
        self.fwd.synchronize()                    # the collective runs on torch's stream, not the engine's
        for k, loc in self._local.items():
            all_gather_samples(loc, self._full[k], self.group)
        torch.cuda.synchronize(self._full["cls"].device)
        self.post.device_raw_pointers(mark_ready=True)
        self.post.posterior(seed=seed, first_image_id=first_image_id)
        self.post.nms()
        self.post.cluster_fuse()
        return [self.post.get_detections(i) for i in range(self.post.B)]


# ---------------------------------------------------------------------------------------------------
# Data-parallel training: every rank runs bod_train_step(apply_update=False) on its own minibatch; the gradients of
# all ~260 tensors sit in ONE contiguous fp32 arena, so the step needs a single all-reduce (39 M floats = 156 MB; a
# ring over 7 xGMI links) instead of per-tensor buckets; the update then runs on the mean gradient (the usual
# data-parallel convention: each rank normalises its loss by its own number of positive anchors).
# ---------------------------------------------------------------------------------------------------
def all_reduce_mean_(grads, group=None):
    """In-place mean of a gradient tensor over the ranks."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return grads
    if grads.is_cuda and dist.get_backend(group) == "gloo":
        host = grads.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        grads.copy_(host)
    else:
        dist.all_reduce(grads, op=dist.ReduceOp.SUM, group=group)
    grads.mul_(1.0 / world)
    return grads


def data_parallel_train_step(engine, images, cls_targets, box_targets, positive_mask, negative_mask, learning_rate, group=None,
                             **step_kwargs):
    """One synchronous data-parallel step; returns this rank's loss dict with the norm of the MEAN gradient."""
    out = engine.train_step(images, cls_targets, box_targets, positive_mask, negative_mask, apply_update=False, **step_kwargs)
    g = engine.train_gradients_view()
    all_reduce_mean_(g, group)
    if g.is_cuda:
        torch.cuda.synchronize(g.device)
    out["grad_norm"] = engine.train_apply(learning_rate)
    return out
