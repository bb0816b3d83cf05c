#!/usr/bin/env python3
"""Experiment (GPU): do two engines driven from two host threads overlap on one GPU -- the memory-bound backbone of one under
the MFMA-bound towers of the other?  Prints frames/s of one engine alone and of two engines concurrently (same batch each).
usage: two_engine_overlap.py [batch per engine] [steps]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from bayes_od_rc_amd import synthetic
from bayes_od_rc_amd.anchor_generator import FpnAnchorGenerator

B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
hw, n = (512, 512), 10
weights = synthetic.make_weights(cls_fg_bias=bench.CALIBRATED_FG_BIAS)
anchors = FpnAnchorGenerator(bench.ANCHOR_CFG).generate_all((hw[0], hw[1], 3))
engines = []
for k in range(2):
    e = bench.make_engine(hw, B, n, 0, precision="bf16", weights=weights, anchors=anchors)
    e.upload_images(synthetic.make_frames(B, hw[0], hw[1], seed=k))
    engines.append(e)

res = {}
def loop(e, k, key=None):
    dt = bench.timed_pipeline(e, k, 2, False, B)        # the bench's own depth-2 pipelined loop
    if key is not None: res[key] = dt

loop(engines[0], 2); loop(engines[1], 2)
loop(engines[0], steps, "solo")
print("one engine, batch %d, pipelined: %.1f frames/s (%.2f ms/step)" % (B, B * steps / res["solo"], res["solo"] / steps * 1e3), flush=True)
ths = [threading.Thread(target=loop, args=(e, steps, i)) for i, e in enumerate(engines)]
t0 = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
t1 = time.perf_counter()
print("two engines concurrently, batch %d each: %.1f frames/s aggregate (wall %.1f ms, threads %.1f / %.1f ms)" % (B, 2 * B * steps / (t1 - t0), (t1 - t0) * 1e3, res[0] * 1e3, res[1] * 1e3), flush=True)
