#!/usr/bin/env python3
"""Two pipelined inference steps of a small engine: the workload of the marker-trace test / profile
(BOD_ROCTX=1 rocprofv3 --marker-trace --kernel-trace -- python3 tests/tools/marker_demo.py [H W B N])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bayes_od_rc_amd import synthetic
from bayes_od_rc_amd.engine import Engine, make_config

H, W, B, N = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (160, 160, 2, 2)))
from bayes_od_rc_amd.anchor_generator import FpnAnchorGenerator
ANCHOR_CFG = {"layers": [3, 4, 5, 6, 7], "aspect_ratios": [[1.0, 1.0], [1.0, 2.0], [2.0, 1.0]],
              "scales": [1.0, 1.26, 1.59], "min_positive_iou": 0.5, "max_negative_iou": 0.4}
eng = Engine(make_config((H, W), batch=B, mc_samples=N, use_full_covar=True,
                         bayes_od_config={"ranking_method": "score", "dirichlet_prior": {"type": "non_informative"},
                                          "gaussian_prior": {"type": "isotropic", "isotropic_variance": 100000.0}},
                         nms_config={"max_output_size": 100, "iou_threshold": 0.5, "soft_nms_sigma": 0.5}))
eng.load_weights(synthetic.make_weights())
eng.set_anchors(FpnAnchorGenerator(ANCHOR_CFG).generate_all((H, W, 3)))
frames = synthetic.make_frames(B, H, W)
eng.upload_images(frames)
slots = []
for step in range(3):
    slots.append(eng.infer_async(None, seed=7 + step, first_image_id=step * B))
    if len(slots) == 2:
        eng.collect(slots.pop(0))
eng.collect(slots.pop(0))
print("marker_demo done")
