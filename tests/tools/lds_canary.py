#!/usr/bin/env python3
"""Run the LDS canary (lds_canary.hip) beside an engine's forward passes: does any kernel of the forward write LDS it does not own?"""
import ctypes as C, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bayes_od_rc_amd import synthetic
from bayes_od_rc_amd.engine import Engine, make_config
HERE = os.path.dirname(os.path.abspath(__file__))
if not os.path.exists(os.path.join(HERE, "liblds_canary.so")):          # (cross-compiles without a GPU; the .so travels with gpurun)
    import subprocess
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(HERE, "lds_canary.hip"), "-o", os.path.join(HERE, "liblds_canary.so")])
lib = C.CDLL(os.path.join(HERE, "liblds_canary.so"))
lib.lds_canary_run.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
hw, n, batch = (512, 512), 2, int(os.environ.get("B", "64"))
prec = os.environ.get("PRECISION", "bf16")
b = Engine(make_config(hw, batch=batch, mc_samples=n, precision=prec)); b.load_weights(synthetic.make_weights()); b.upload_images(synthetic.make_frames(batch, hw[0], hw[1], seed=12))
b.forward(None)
out = (C.c_uint32 * 8)()
words = int(os.environ.get("WORDS", "2048"))
print("alone:", lib.lds_canary_run(4096, words, 200, out), list(out)[:5], flush=True)
stop = False
def noise():
    while not stop: b.forward(None, seed=1, first_image_id=0)
t = threading.Thread(target=noise); t.start()
try:
    for it in range(10):
        rc = lib.lds_canary_run(4096, words, 2000, out)
        print("beside the forward (%s): rc %d mismatches %d first word %d value 0x%08x block %d iteration %d" % (prec, rc, out[0], out[1], out[2], out[3], out[4]), flush=True)
finally:
    stop = True; t.join()
