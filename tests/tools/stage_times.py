#!/usr/bin/env python3
"""Where a bench step's time goes outside the convolutions: times the same B frames through
(a) forward only, (b) forward + posterior, (c) + NMS + cluster-fuse on the main stream (bod_infer, no
pipelining), (d) the pipelined infer_async/collect loop bench.py uses.  Development aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench as B
from bayes_od_rc_amd import synthetic
from bayes_od_rc_amd.anchor_generator import FpnAnchorGenerator
from bayes_od_rc_amd.engine import Engine, make_config

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = 8
hw = (512, 512)
eng = Engine(make_config(hw, batch=batch, mc_samples=10, bayes_od_config=B.BAYES_CFG, nms_config=B.NMS_CFG, use_full_covar=True))
eng.load_weights(synthetic.make_weights(cls_fg_bias=B.CALIBRATED_FG_BIAS))
eng.set_anchors(FpnAnchorGenerator(B.ANCHOR_CFG).generate_all((hw[0], hw[1], 3)))
eng.upload_images(synthetic.make_frames(batch, hw[0], hw[1], seed=0))

def timeit(fn, label):
    for i in range(2): fn(i)
    eng.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): fn(i)
    eng.synchronize()
    print("%-40s %8.3f ms/step" % (label, (time.perf_counter() - t0) / steps * 1e3), flush=True)

timeit(lambda i: eng.forward(None, 0, i * batch), "forward")
def fp(i):
    eng.forward(None, 0, i * batch); eng.posterior(0, i * batch)
timeit(fp, "forward+posterior")
def fpn(i):
    eng.forward(None, 0, i * batch); eng.posterior(0, i * batch); eng.nms()
timeit(fpn, "forward+posterior+nms")
timeit(lambda i: eng.infer(None, 0, i * batch), "infer (all stages, main stream)")
pend, out = [], [None, None]
def pipe(i):
    pend.append(eng.infer_async(None, 0, i * batch))
    if len(pend) > 1:
        s = pend.pop(0); out[s] = eng.collect(s, out[s])
def run_pipe(label):
    for i in range(2): pipe(i)
    while pend:
        s = pend.pop(0); out[s] = eng.collect(s, out[s])
    eng.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): pipe(i)
    while pend:
        s = pend.pop(0); out[s] = eng.collect(s, out[s])
    eng.synchronize()
    print("%-40s %8.3f ms/step" % (label, (time.perf_counter() - t0) / steps * 1e3), flush=True)
run_pipe("pipelined infer_async/collect")
