#!/usr/bin/env python3
"""Worker of tests/test_gpu_pipeline.py::test_sample_sharded_two_ranks_on_one_gpu: every rank (all on GPU 0, gloo)
runs its share of the MC ensemble through distributed.SampleShardedEngine and rank 0 compares the detections with a
single handle running the whole ensemble."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.distributed as dist
from bayes_od_rc_amd import synthetic, distributed as bd
from bayes_od_rc_amd.anchor_generator import FpnAnchorGenerator
from bayes_od_rc_amd.engine import Engine, make_config

ACFG = {"layers": [3, 4, 5, 6, 7], "aspect_ratios": [[1.0, 1.0], [1.0, 2.0], [2.0, 1.0]], "scales": [1.0, 1.26, 1.59]}
dist.init_process_group("gloo")
rank = dist.get_rank()
torch.cuda.set_device(0)
hw, n_total = (160, 160), 6
weights = synthetic.make_weights(cls_fg_bias=-1.0)
anchors = FpnAnchorGenerator(ACFG).generate_all((hw[0], hw[1], 3))
frames = synthetic.make_frames(1, hw[0], hw[1], seed=9)
sse = bd.SampleShardedEngine(hw, weights, anchors, n_total, use_full_covar=True)
got = sse.infer(frames, seed=1, first_image_id=7)
ok = True
if rank == 0:
    ref = Engine(make_config(hw, batch=1, mc_samples=n_total, use_full_covar=True))
    ref.load_weights(weights)
    ref.set_anchors(anchors)
    ref.infer(frames, seed=1, first_image_id=7)
    ok = all(np.array_equal(a, b) for a, b in zip(got[0], ref.get_detections(0))) and got[0][0].shape[0] > 0
    print("SAMPLE_SHARD_OK" if ok else "SAMPLE_SHARD_MISMATCH", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
