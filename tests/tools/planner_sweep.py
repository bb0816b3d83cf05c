#!/usr/bin/env python3
"""Sweep (GPU): the forward plan's kernel choices against their alternatives, by batch and geometry.
For B in {1, 3, 8, 32, 128, 256} x {512x512, 384x1248}: median-of-5 wall time (engine synchronised on both sides, several forwards per sample) of the N=1 forward (backbone + FPN + one-sample heads:
the launches the streaming kernels -- sliding-window 3x3, pointwise 1x1, fused stem + pool -- compete for) under the default plan and
with each of those kernels switched off (BOD_SLIDE3X3=0 / BOD_POINTWISE=0 / BOD_STEM_POOL_FUSED=0) and, since round 4, with each of that
round's plan changes undone (BOD_PLANE_XREUSE=0 / BOD_SLIDE3X3_C128=0 / BOD_COUT_INNER=0 / BOD_STEM_WAVES=4).  Every configuration runs in a child
process (the switches are read once per process).  usage: planner_sweep.py [--json]   (tests/test_gpu_planner.py asserts on it)"""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SWITCHES = {"default": {}, "no_slide3x3": {"BOD_SLIDE3X3": "0"}, "no_pointwise": {"BOD_POINTWISE": "0"},
            "no_fused_stem_pool": {"BOD_STEM_POOL_FUSED": "0"},
            # round 4: 256 -> 256 3x3 layers on the tower loop, stage 3's 128-channel sliding window, cout tiles side by side on an XCD,
            # the eight-wave stem + pool kernel
            "no_plane_row_reuse": {"BOD_PLANE_XREUSE": "0"}, "no_slide3x3_c128": {"BOD_SLIDE3X3_C128": "0"},
            "no_cout_inner": {"BOD_COUT_INNER": "0"}, "stem_four_waves": {"BOD_STEM_WAVES": "4"}}
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
from bayes_od_rc_amd import synthetic
from bayes_od_rc_amd.engine import Engine, make_config
hw, B, reps = (%d, %d), %d, %d
eng = Engine(make_config(hw, batch=B, mc_samples=1))
eng.load_weights(synthetic.make_weights())
eng.upload_images(synthetic.make_frames(B, hw[0], hw[1], seed=1))
for _ in range(3): eng.forward(None)
eng.synchronize()
ts = []
inner = max(1, min(20, 2048 // max(B, 1) // 8))
import time
for r in range(reps):
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(inner): eng.forward(None)
    eng.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3 / inner)
ts.sort()
print(json.dumps({"ms": ts[len(ts) // 2], "min": ts[0], "max": ts[-1]}))
'''


def measure(hw, B, env, reps=5):
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, "-c", CHILD % (ROOT, hw[0], hw[1], B, reps)], env=e, capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        raise RuntimeError(out.stderr[-2000:])
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def sweep(batches=(1, 3, 8, 32, 128, 256), geoms=((512, 512), (384, 1248)), names=None):
    rows = []
    for hw in geoms:
        for B in batches:
            row = {"hw": list(hw), "batch": B}
            for name, env in SWITCHES.items():
                if names and name not in names:
                    continue
                row[name] = measure(hw, B, env)["ms"]
            rows.append(row)
            print("%dx%d B=%3d  " % (hw[0], hw[1], B) + "  ".join("%s %.3f" % (k, v) for k, v in row.items() if k not in ("hw", "batch")),
                  file=sys.stderr, flush=True)
    return rows


if __name__ == "__main__":
    rows = sweep()
    if "--json" in sys.argv:
        print(json.dumps(rows))
