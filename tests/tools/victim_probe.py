#!/usr/bin/env python3
"""Do OTHER kernels of the process compute wrong results while an engine's forward runs beside them?  A torch elementwise kernel
(16-byte vector loads and stores) and a copy, checked against their results without company.  (development probe)"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bayes_od_rc_amd import synthetic
from bayes_od_rc_amd.engine import Engine, make_config
hw, n, batch = (512, 512), 2, int(os.environ.get("B", "64"))
prec = os.environ.get("PRECISION", "bf16")
b = Engine(make_config(hw, batch=batch, mc_samples=n, precision=prec)); b.load_weights(synthetic.make_weights()); b.upload_images(synthetic.make_frames(batch, hw[0], hw[1], seed=12))
b.forward(None)
x = torch.randn(32 * 1024 * 1024, device="cuda")
ref = (x * 2 + 1); refc = x.clone(); torch.cuda.synchronize()
stop = False
def noise():
    while not stop: b.forward(None, seed=1, first_image_id=0)
for company in (False, True):
    if company:
        t = threading.Thread(target=noise); t.start()
    bad = 0; badc = 0; first = None
    for it in range(200):
        y = x * 2 + 1; c = x.clone(); torch.cuda.synchronize()
        if not torch.equal(y, ref):
            bad += 1
            if first is None:
                d = torch.nonzero(y != ref).flatten()
                first = (int(d.numel()), d[:20].tolist())
        if not torch.equal(c, refc): badc += 1
    print("company %s: %d of 200 elementwise results wrong, %d of 200 copies wrong; first: %s" % (company, bad, badc, first), flush=True)
stop = True
if company: t.join()
