#!/usr/bin/env python3
"""f16mx4 (fp4 cross terms) against f16mx and float64 on single tower layers and one small end-to-end forward (development check)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bayes_od_rc_amd.engine import stage_conv
from oracle import network, philox


def rel_err(a, t, floor):
    a = np.asarray(a, np.float64); t = np.asarray(t, np.float64)
    return float((np.abs(a - t) / (np.abs(t) + floor)).max())


rng = np.random.default_rng(3)
for (b, h, w) in ((2, 16, 16), (1, 21, 37)):
    x = np.maximum(rng.normal(0, 1, (b, h, w, 256)), 0).astype(np.float32) * (rng.random((b, h, w, 256)) >= 0.3).astype(np.float32) / np.float32(0.7)
    wt = (rng.normal(0, 1, (3, 3, 256, 256)) * np.sqrt(2.0 / (9 * 256))).astype(np.float32)
    bias = rng.normal(0, 0.5, 256).astype(np.float32)
    ref = np.maximum(network.conv2d(x.astype(np.float64), wt.astype(np.float64), bias.astype(np.float64), 1, "same"), 0)
    rms = float(np.sqrt((ref ** 2).mean()))
    for prec in ("f16mx", "f16mx4"):
        for mode in (0, 1, 2):
            got = stage_conv(x, wt, bias, padding="same", relu=True, precision=prec, round_output_bf16=mode)
            d = got - ref
            print("%s mode %d %dx%dx%d: max %.2e  rms %.2e  (nan %d)" % (prec, mode, b, h, w, rel_err(got, ref, rms), float(np.sqrt((d ** 2).mean())) / rms, int(np.isnan(got).sum())), flush=True)
    keep = np.stack([philox.dropout_keep_mask(11, 5, s, 6, h * w, 256, 0.3).reshape(h, w, 256) for s in range(b)])
    refd = ref * np.float64(np.float32(1.0 / 0.7)) * keep
    for mode in (0, 1, 2):
        got = stage_conv(x, wt, bias, padding="same", relu=True, dropout_rate=0.3, seed=11, layer_id=6, image_id=5, precision="f16mx4", round_output_bf16=mode)
        print("f16mx4 dropout mode %d: max %.2e zeros ok %s" % (mode, rel_err(got, refd, float(np.sqrt((refd ** 2).mean()))), bool(np.all(got[~keep] == 0))), flush=True)
