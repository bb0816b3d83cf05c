#!/usr/bin/env python3
"""Run the gather canary (gather_canary.hip) alone and beside an engine's forward passes: do gathered 16-byte / 8-byte loads change?"""
import ctypes as C, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
if not os.path.exists(os.path.join(HERE, "libgather_canary.so")):
    import subprocess
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(HERE, "gather_canary.hip"), "-o", os.path.join(HERE, "libgather_canary.so")])
from bayes_od_rc_amd import synthetic
from bayes_od_rc_amd.engine import Engine, make_config
lib = C.CDLL(os.path.join(HERE, "libgather_canary.so"))
lib.gather_canary_run.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_float)]
hw, n, batch = (512, 512), 2, int(os.environ.get("B", "64"))
b = Engine(make_config(hw, batch=batch, mc_samples=n, precision=os.environ.get("PRECISION", "bf16"))); b.load_weights(synthetic.make_weights()); b.upload_images(synthetic.make_frames(batch, hw[0], hw[1], seed=12))
b.forward(None)
BLOCKS, ITERS, RUNS = 8192, 64, int(os.environ.get("RUNS", "100"))
def run():
    out = np.empty(BLOCKS * 256, np.float32)
    rc = lib.gather_canary_run(BLOCKS, ITERS, out.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc == 0, rc
    return out
ref = run()
assert np.array_equal(run().view(np.uint32), ref.view(np.uint32)), "not reproducible alone"
stop = False
def company():
    while not stop: b.forward(None, seed=1, first_image_id=0)
for name, fn in (("alone", None), ("this library's forward", company)):
    stop = False
    t = threading.Thread(target=fn) if fn else None
    if t: t.start()
    try:
        bad, rows = 0, []
        for r in range(RUNS):
            out = run()
            d = np.nonzero(out.view(np.uint32) != ref.view(np.uint32))[0]
            if len(d):
                bad += 1; rows.append((len(d), int(d[0]), int(d[-1]) - int(d[0]) + 1, int(d[0]) % 16))
        print("%-24s: %d of %d runs differ from the run alone; (lanes wrong, first lane, span, first lane %% 16) %s" % (name, bad, RUNS, rows[:6]), flush=True)
    finally:
        stop = True
        if t: t.join()
