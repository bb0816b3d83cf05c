#!/usr/bin/env python3
"""Serial handle: two calls in flight (infer_async x 2) against the synchronous call, repeated on fresh handles.  (development check)"""
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
hw = tuple(int(v) for v in os.environ.get("RACE_HW", "128x128").split("x"))
n, batch = 2, int(os.environ.get("RACE_BATCH", "128"))
weights = synthetic.make_weights(cls_fg_bias=-1.0)
anchors = FpnAnchorGenerator(ANCHOR_CFG).generate_all((hw[0], hw[1], 3))
frames = synthetic.make_frames(batch, hw[0], hw[1], seed=12)
def mk():
    e = Engine(make_config(hw, batch=batch, mc_samples=n, bayes_od_config=BAYES_CFG, nms_config=NMS_CFG, use_full_covar=True))
    e.load_weights(weights); e.set_anchors(anchors); e.upload_images(frames)
    return e
def diff(a, b, tag):
    nbad = 0
    for img in range(batch):
        k = a["num"][img]
        for key in ("scores", "means", "covs", "counts"):
            if a["num"][img] != b["num"][img] or not np.array_equal(a[key][img, :k], b[key][img, :k]):
                nbad += 1
                if nbad <= 4: print("%s: img %d %s differs, max |d| %.3g" % (tag, img, key, float(np.abs(a[key][img, :k] - b[key][img, :k]).max())), flush=True)
    return nbad
order = os.environ.get("ORDER", "async_first")
res = {}
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    e = mk()
    if order == "sync_first" or it > 0:
        e.infer(None, seed=3, first_image_id=0)
        res.setdefault("sync", {k: v.copy() for k, v in e.get_detections_batch().items()})
        print("iter %d sync vs first sync: %d arrays differ" % (it, diff(res["sync"], e.get_detections_batch(), "sync")), flush=True)
        e.close(); e = mk()
    s0 = e.infer_async(None, seed=3, first_image_id=0); s1 = e.infer_async(None, seed=3, first_image_id=batch)
    d0 = {k: v.copy() for k, v in e.collect(s0).items()}; e.collect(s1)
    res.setdefault("async", d0)
    if "sync" in res: print("iter %d pipelined batch 0 vs sync: %d arrays differ" % (it, diff(res["sync"], d0, "pipelined vs sync")), flush=True)
    print("iter %d pipelined batch 0 vs first pipelined: %d arrays differ" % (it, diff(res["async"], d0, "pipelined vs first pipelined")), flush=True)
    e.close()
