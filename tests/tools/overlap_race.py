#!/usr/bin/env python3
"""Which handle is not reproducible at 64 frames: the serial pipeline or the CU-partitioned overlapped one?  (development check)"""
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
hw = tuple(int(v) for v in os.environ.get("RACE_HW", "512x512").split("x"))
n, batch = 2, int(os.environ.get("RACE_BATCH", "64"))
weights = synthetic.make_weights(cls_fg_bias=-1.0)
anchors = FpnAnchorGenerator(ANCHOR_CFG).generate_all((hw[0], hw[1], 3))
frames = synthetic.make_frames(batch, hw[0], hw[1], seed=12)
ref = None
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for it in range(iters):
    for overlap in (False, True):
        if overlap:
            os.environ["BOD_OVERLAP"] = os.environ.get("RACE_OVERLAP", "1")
        e = Engine(make_config(hw, batch=batch, mc_samples=n, bayes_od_config=BAYES_CFG, nms_config=NMS_CFG, use_full_covar=True, pipeline_overlap=overlap))
        os.environ.pop("BOD_OVERLAP", None)
        e.load_weights(weights); e.set_anchors(anchors); e.upload_images(frames)
        s0 = e.infer_async(None, seed=3, first_image_id=0)
        s1 = e.infer_async(None, seed=3, first_image_id=batch)
        d = [{k: v.copy() for k, v in e.collect(s).items()} for s in (s0, s1)]
        pyr = e.get_pyramid(2).copy()
        e.close()
        if ref is None:
            ref = (d, pyr)
            continue
        for bi in (0, 1):
            a, b = ref[0][bi], d[bi]
            for img in range(batch):
                k = a["num"][img]
                for key in ("scores", "means", "covs", "counts"):
                    if a["num"][img] != b["num"][img] or not np.array_equal(a[key][img, :k], b[key][img, :k]):
                        dd = np.argwhere(a[key][img, :k] != b[key][img, :k])
                        print("iter %d overlap %s batch %d img %d %s: %d elements differ, dets %s, max |d| %.3g" % (
                            it, overlap, bi, img, key, len(dd), sorted(set(int(x[0]) for x in dd))[:8], float(np.abs(a[key][img, :k] - b[key][img, :k]).max())), flush=True)
        if not np.array_equal(ref[1], pyr):
            print("iter %d overlap %s: pyramid differs" % (it, overlap), flush=True)
print("done", flush=True)
