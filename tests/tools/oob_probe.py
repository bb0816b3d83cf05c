#!/usr/bin/env python3
"""Does one handle's forward write into ANOTHER handle's buffers?  (development check: out-of-bounds stores)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from conftest import ANCHOR_CFG
from bayes_od_rc_amd import synthetic
from bayes_od_rc_amd.anchor_generator import FpnAnchorGenerator
from bayes_od_rc_amd.engine import Engine, make_config

BAYES_CFG = {"ranking_method": "score", "dirichlet_prior": {"type": "non_informative"}, "gaussian_prior": {"type": "isotropic", "isotropic_variance": 100000.0}}
NMS_CFG = {"max_output_size": 100, "iou_threshold": 0.5, "soft_nms_sigma": 0.5}
hw, n, batch = (512, 512), 2, 64
weights = synthetic.make_weights(cls_fg_bias=-1.0)
anchors = FpnAnchorGenerator(ANCHOR_CFG).generate_all((hw[0], hw[1], 3))
frames = synthetic.make_frames(batch, hw[0], hw[1], seed=12)
mk = lambda: Engine(make_config(hw, batch=batch, mc_samples=n, bayes_od_config=BAYES_CFG, nms_config=NMS_CFG, use_full_covar=True))
e = mk(); e.load_weights(weights); e.set_anchors(anchors); e.upload_images(frames)
b = mk(); b.load_weights(weights); b.set_anchors(anchors); b.upload_images(frames)
e.infer(None, seed=3, first_image_id=0)
snap = lambda: ([e.get_posterior(i) for i in range(batch)], {k: v.copy() for k, v in e.get_detections_batch().items()}, [e.get_pyramid(l).copy() for l in range(5)])
p0 = snap()
for it in range(4):
    b.forward(None, seed=1, first_image_id=0) if it % 2 == 0 else b.infer(None, seed=1, first_image_id=0)
    b.synchronize()
    p1 = snap()
    bad = 0
    for i in range(batch):
        for k in p0[0][i]:
            if not np.array_equal(p0[0][i][k], p1[0][i][k]):
                bad += 1
                d = np.argwhere(np.asarray(p0[0][i][k]) != np.asarray(p1[0][i][k]))
                print("after b.%s #%d: e's posterior img %d %s changed at %d places, first %s" % ("forward" if it % 2 == 0 else "infer", it, i, k, len(d), d[:3].tolist()), flush=True)
    for k in p0[1]:
        if not np.array_equal(p0[1][k], p1[1][k]): bad += 1; print("after #%d: e's detections %s changed" % (it, k), flush=True)
    for l in range(5):
        if not np.array_equal(p0[2][l], p1[2][l]): bad += 1; print("after #%d: e's pyramid level %d changed" % (it, l), flush=True)
    print("iteration %d: %d arrays of handle e changed" % (it, bad), flush=True)
    e.posterior(seed=3, first_image_id=0); e.nms(); e.cluster_fuse()
    p2 = snap()
    bad = sum(1 for i in range(batch) for k in p0[0][i] if not np.array_equal(p0[0][i][k], p2[0][i][k])) + sum(1 for k in p0[1] if not np.array_equal(p0[1][k], p2[1][k]))
    print("iteration %d: e's posterior RE-RUN (no concurrency): %d arrays differ from the first run" % (it, bad), flush=True)
