#!/usr/bin/env python3
"""Worker of tests/test_gpu_train_step.py::test_data_parallel_two_ranks_on_one_gpu: two ranks (both on GPU 0, gloo) train
on different minibatches with ONE all-reduce of the gradient arena per step.  Checked: after the first step the weights
equal what a single process gets by averaging the two ranks' gradient arenas by hand, and after three steps
both ranks still hold identical weights."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.distributed as dist
from bayes_od_rc_amd import synthetic, constants, distributed as bd
from bayes_od_rc_amd.engine import Engine, make_config
from bayes_od_rc_amd.run_training import synthetic_samples

ACFG = {'layers': [3, 4, 5, 6, 7], 'aspect_ratios': [[1, 1], [1, 2], [2, 1]], 'scales': [1.0, 1.26, 1.59], 'min_positive_iou': 0.5, 'max_negative_iou': 0.4}
PROBE = (("pyramid_classification_1", "kernel", (3, 3, 256, 256)), ("res3b_branch2b", "kernel", (3, 3, 128, 128)), ("bn_conv1", "gamma", (64,)))
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
hw, batch = (96, 96), 2
weights = synthetic.make_weights(cls_fg_bias=-2.0)


def batch_of(r):
    s = synthetic_samples(batch, hw, ACFG, 7, seed=100 + r)
    st = lambda k: np.stack([x[k] for x in s])
    data = (st(constants.IMAGE_NORMALIZED_KEY), st(constants.ANCHORS_CLASS_TARGETS_KEY), st(constants.ANCHORS_BOX_TARGETS_KEY),
            st(constants.POSITIVE_ANCHORS_MASK_KEY), st(constants.NEGATIVE_ANCHOR_MASK_KEY))
    return data, np.asarray(s[0][constants.ANCHORS_KEY], np.float32)


def new_engine(anchors):
    e = Engine(make_config(hw, batch=batch, mc_samples=1, training=True))
    e.load_weights(weights)
    e.set_anchors(anchors)
    return e


def probe(e):
    return [e.train_get(*p) for p in PROBE]

mine, anchors = batch_of(rank)
eng = new_engine(anchors)
out = bd.data_parallel_train_step(eng, *mine, learning_rate=1e-3, seed=5, first_image_id=rank * batch)
after_one = probe(eng)
ok = np.isfinite(out["total_loss"])
if rank == 0:
    refs, views = [], []
    for r in range(world):
        e = new_engine(anchors)
        e.train_step(*batch_of(r)[0], apply_update=False, seed=5, first_image_id=r * batch)
        refs.append(e)
        views.append(e.train_gradients_view())
    mean = (views[0] + views[1]) * 0.5
    views[0].copy_(mean)
    torch.cuda.synchronize()
    refs[0].train_apply(1e-3)
    # (two runs of a step agree to the last few bits only: fp32 atomics in the backward pass)
    diffs = [float(np.abs(a - b).max()) for a, b in zip(after_one, probe(refs[0]))]
    print("dp vs hand-averaged reference, max |dw|:", diffs, flush=True)
    ok = ok and all(d <= 1e-6 for d in diffs)
    changed = any(not np.array_equal(a, np.asarray(weights[p[0]][p[1]], np.float32)) for a, p in zip(after_one, PROBE))
    print("weights changed:", changed, flush=True)
    ok = ok and changed
for step in (1, 2):
    out = bd.data_parallel_train_step(eng, *mine, learning_rate=1e-3, seed=5, first_image_id=(step * world + rank) * batch)
    ok = ok and np.isfinite(out["total_loss"])
for a in probe(eng):
    t = torch.from_numpy(a.copy())
    both = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(both, t)
    same = all(torch.equal(both[0], b) for b in both)
    if rank == 0 and not same:
        print("ranks differ by", float((both[0] - both[1]).abs().max()), flush=True)
    ok = ok and same
flag = torch.tensor([1 if ok else 0])
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
if rank == 0:
    print("DP_TRAIN_OK" if int(flag) == 1 else "DP_TRAIN_MISMATCH", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if int(flag) == 1 else 1)
