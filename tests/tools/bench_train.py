#!/usr/bin/env python3
"""Training-step throughput (BASELINE config 5's step on the ResNet-50 model): steps/s and frames/s of bod_train_step
at the yaml's minibatch (3) and at 8 frames, 512x512, full-covariance loss.  usage: bench_train.py [H W] [batch] [depth]
(depth 101 = BASELINE config 5's "ResNet-101")"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bayes_od_rc_amd import synthetic
from bayes_od_rc_amd.engine import Engine, make_config
from bayes_od_rc_amd.run_training import synthetic_samples
from bayes_od_rc_amd import constants

hw = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 512)
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 50
for batch in ([int(sys.argv[3])] if len(sys.argv) > 3 else [3, 8]):
    acfg = {'layers': [3, 4, 5, 6, 7], 'aspect_ratios': [[1, 1], [1, 2], [2, 1]], 'scales': [1.0, 1.26, 1.59], 'min_positive_iou': 0.5, 'max_negative_iou': 0.4}
    samples = synthetic_samples(batch, hw, acfg, 7)
    eng = Engine(make_config(hw, batch=batch, mc_samples=1, training=True, backbone_depth=depth))
    eng.load_weights(synthetic.make_weights(depth=depth))
    eng.set_anchors(np.asarray(samples[0][constants.ANCHORS_KEY], np.float32))
    st = lambda k: np.stack([s[k] for s in samples])
    args = (st(constants.IMAGE_NORMALIZED_KEY), st(constants.ANCHORS_CLASS_TARGETS_KEY), st(constants.ANCHORS_BOX_TARGETS_KEY),
            st(constants.POSITIVE_ANCHORS_MASK_KEY), st(constants.NEGATIVE_ANCHOR_MASK_KEY))
    eng.upload_images(args[0])
    for i in range(3):
        out = eng.train_step(None, *args[1:], seed=1, first_image_id=i * batch)
    t0 = time.perf_counter()
    n = 10
    for i in range(n):
        out = eng.train_step(None, *args[1:], seed=1, first_image_id=(3 + i) * batch)
    dt = (time.perf_counter() - t0) / n
    print("train step ResNet-%d %dx%d batch %d: %.1f ms/step, %.1f frames/s, loss %.3f, device bytes %.2f GB" % (depth, hw[0], hw[1], batch, dt * 1e3, batch / dt, out["total_loss"], eng.device_bytes / 1e9), flush=True)
    eng.close()
