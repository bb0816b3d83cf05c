#!/usr/bin/env python3
"""Micro-benchmark of the head-tower conv kernel on the bench geometry (B=8, N=10, 512x512).
usage: bench_head_conv.py [variant[:layer]...]   e.g.  bench_head_conv.py 0 0:2 0:3 90
(variant 0 = production, 90 = phase clock, 1 / 2 / 4 / 30 / 31 = ablation builds, see conv_igemm.hip)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bayes_od_rc_amd import synthetic
from bayes_od_rc_amd.engine import Engine, make_config

variants = [(int(v.split(":")[0]), int(v.split(":")[1]) if ":" in v else None) for v in sys.argv[1:]] or [(0, None)]
B = int(os.environ.get("B", "8"))
eng = Engine(make_config((512, 512), batch=B, mc_samples=10, precision=os.environ.get("PRECISION", "bf16")))
eng.load_weights(synthetic.make_weights())
eng.upload_images(synthetic.make_frames(B, 512, 512))
eng.forward(None)          # real (random-data) activations in the buffers
eng.synchronize()
for rnd in range(2):
    for v, want in variants:
        for layer in ((want,) if want is not None else (1, 0) if v == 0 else (0,) if 91 <= v <= 95 else (1,)):
            ms, fl = eng.bench_head_conv(layer=layer, variant=v, iters=10)
            print("round %d variant %d layer %d: %.3f ms  %.1f TFLOP/s (%.1f%% of 2500)" % (rnd, v, layer, ms, fl / ms / 1e9, fl / ms / 1e9 / 25))
