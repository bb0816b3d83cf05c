#!/usr/bin/env python3
"""Run the VALU canary (valu_canary.hip) alone, beside torch GEMMs and beside an engine's forward passes: whose company changes its results?"""
import ctypes as C, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
if not os.path.exists(os.path.join(HERE, "libvalu_canary.so")):
    import subprocess
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(HERE, "valu_canary.hip"), "-o", os.path.join(HERE, "libvalu_canary.so")])
from bayes_od_rc_amd import synthetic
from bayes_od_rc_amd.engine import Engine, make_config
lib = C.CDLL(os.path.join(HERE, "libvalu_canary.so"))
lib.valu_canary_run.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
hw, n, batch = (512, 512), 2, int(os.environ.get("B", "64"))
b = Engine(make_config(hw, batch=batch, mc_samples=n, precision=os.environ.get("PRECISION", "bf16"))); b.load_weights(synthetic.make_weights()); b.upload_images(synthetic.make_frames(batch, hw[0], hw[1], seed=12))
b.forward(None)
BLOCKS, ITERS, RUNS = 8192, 3000, int(os.environ.get("RUNS", "40"))
def run(mode):
    out = np.empty(BLOCKS * 256, np.float32)
    rc = lib.valu_canary_run(BLOCKS, ITERS, mode, out.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc == 0, rc
    return out
NM = 7
refs = [run(m) for m in range(NM)]
assert all(np.array_equal(run(m).view(np.uint32), refs[m].view(np.uint32)) for m in range(NM)), "not reproducible alone"
stop = False
def company_forward():
    while not stop: b.forward(None, seed=1, first_image_id=0)
def company_torch():
    import torch
    x = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16); y = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    while not stop:
        for _ in range(8): z = x @ y
        torch.cuda.synchronize()
for name, fn in (("alone", None), ("this library's forward", company_forward)):
    stop = False
    t = threading.Thread(target=fn) if fn else None
    if t: t.start()
    try:
        for mode, mname in enumerate(("fma chain", "sqrt / rcp / exp chain", "IEEE division chain", "integer multiply chain", "fp64 fma chain", "packed fp32 chain", "64-bit shift-add chain")):
            bad_runs, rows = 0, []
            for r in range(RUNS):
                out = run(mode)
                d = np.nonzero(out.view(np.uint32) != refs[mode].view(np.uint32))[0]
                if len(d):
                    bad_runs += 1
                    rows.append((len(d), int(d[0]), int(d[-1]) - int(d[0]) + 1, int(d[0]) % 16))
            print("%-24s %-24s: %d of %d runs differ from the run alone; (lanes wrong, first lane, span, first lane %% 16) %s" % (name, mname, bad_runs, RUNS, rows[:4]), flush=True)
    finally:
        stop = True
        if t: t.join()
