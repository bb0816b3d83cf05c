#!/usr/bin/env python3
"""Which stage of posterior -> soft-NMS -> cluster-and-fuse is not reproducible while another handle keeps the GPU busy?  (development check)"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from conftest import ANCHOR_CFG
from bayes_od_rc_amd import synthetic
from bayes_od_rc_amd.anchor_generator import FpnAnchorGenerator
from bayes_od_rc_amd.engine import Engine, make_config

BAYES_CFG = {"ranking_method": "score", "dirichlet_prior": {"type": "non_informative"}, "gaussian_prior": {"type": "isotropic", "isotropic_variance": 100000.0}}
NMS_CFG = {"max_output_size": 100, "iou_threshold": 0.5, "soft_nms_sigma": 0.5}
hw, n, batch = (512, 512), 2, 64
weights = synthetic.make_weights(cls_fg_bias=float(os.environ.get("FG_BIAS", "-1.0")))
anchors = FpnAnchorGenerator(ANCHOR_CFG).generate_all((hw[0], hw[1], 3))
frames = synthetic.make_frames(batch, hw[0], hw[1], seed=12)
mk = lambda prec="bf16": Engine(make_config(hw, batch=batch, mc_samples=n, bayes_od_config=BAYES_CFG, nms_config=NMS_CFG, use_full_covar=True, precision=prec))
e = mk(); e.load_weights(weights); e.set_anchors(anchors); e.upload_images(frames)
b = mk(os.environ.get("NOISE_PRECISION", "bf16")); b.load_weights(weights); b.set_anchors(anchors)
b.upload_images(frames if os.environ.get("NOISE_FRAMES", "same") == "same" else synthetic.make_frames(batch, hw[0], hw[1], seed=977)[:, ::-1].copy()); b.forward(None)
e.infer(None, seed=3, first_image_id=0)
ref = {k: v.copy() for k, v in e.get_detections_batch().items()}
ref_post = [e.get_posterior(i) for i in range(batch)]
ref_nms = [e.get_nms(i) for i in range(batch)]
stop = False
def noise():
    kind = os.environ.get("NOISE_KIND", "forward")
    if kind == "torch":                       # someone else's kernels: rocBLAS GEMMs + elementwise kernels on torch's own stream
        import torch
        x = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16); y = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
        while not stop:
            z = (x @ y).relu_().float().sum(); torch.cuda.synchronize()
        return
    if kind.startswith("synthetic"):          # one instruction class at a time (noise_kernels.hip: mode after the colon)
        import ctypes as C
        here = os.path.dirname(os.path.abspath(__file__))
        if not os.path.exists(os.path.join(here, "libnoise_kernels.so")):
            import subprocess
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(here, "noise_kernels.hip"), "-o", os.path.join(here, "libnoise_kernels.so")])
        lib = C.CDLL(os.path.join(here, "libnoise_kernels.so"))
        mode = int(kind.split(":")[1])
        while not stop:
            assert lib.noise_run(mode, 8192, 2000 if mode != 3 else 400) == 0
        return
    if kind.startswith("tower"):              # one tower launch over and over (bench_head_conv: layer given after the colon)
        while not stop: b.bench_head_conv(layer=int(kind.split(":")[1]), variant=0, iters=20)
        return
    while not stop:
        b.forward(None, seed=1, first_image_id=0)
NOISE = os.environ.get("NOISE", "1") != "0"
t = threading.Thread(target=noise if NOISE else (lambda: None)); t.start()
def cmp_dets(tag):
    d = e.get_detections_batch()
    bad = 0
    for img in range(batch):
        k = ref["num"][img]
        for key in ("scores", "means", "covs", "counts"):
            if d["num"][img] != k or not np.array_equal(d[key][img, :k], ref[key][img, :k]):
                bad += 1
                print("%s: img %d %s differs (max |d| %.3g)" % (tag, img, key, float(np.abs(d[key][img, :k] - ref[key][img, :k]).max())), flush=True)
    return bad
try:
    tot = {"cluster": 0, "nms+cluster": 0, "posterior+nms+cluster": 0, "posterior_arrays": 0, "nms_lists": 0}
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
        e.cluster_fuse(); tot["cluster"] += cmp_dets("iter %d cluster only" % it)
        e.nms(); e.cluster_fuse(); tot["nms+cluster"] += cmp_dets("iter %d nms+cluster" % it)
        for i in range(batch):
            a, r = e.get_nms(i), ref_nms[i]
            if not all(np.array_equal(x, y) for x, y in zip(a, r)): tot["nms_lists"] += 1
        e.posterior(seed=3, first_image_id=0)
        for i in range(0, batch, 1):
            a, r = e.get_posterior(i), ref_post[i]
            for k in r:
                if not np.array_equal(a[k], r[k]):
                    tot["posterior_arrays"] += 1
                    if tot["posterior_arrays"] <= 12:
                        d = np.argwhere(np.asarray(a[k]) != np.asarray(r[k]))
                        rows = sorted(set(int(x[0]) for x in d))
                        print("iter %d posterior img %d %s differs in %d slots %s of %d; anchor_index equal %s; first slot got %s ref %s" % (
                            it, i, k, len(rows), rows[:6], len(r[k]), np.array_equal(a["anchor_index"], r["anchor_index"]),
                            np.asarray(a[k])[rows[0]].ravel()[:4], np.asarray(r[k])[rows[0]].ravel()[:4]), flush=True)
                        if k == "means":          # are the wrong rows the RIGHT rows of other slots (of this image or any other)?
                            for rr in rows[:16]:
                                hit = [(j, int(np.nonzero((ref_post[j]["means"] == a[k][rr]).all(axis=1))[0][0])) for j in range(batch)
                                       if (ref_post[j]["means"] == a[k][rr]).all(axis=1).any()]
                                print("      slot %d: its wrong mean is the right mean of (image, slot) %s; anchor %d" % (rr, hit[:3], int(r["anchor_index"][rr])), flush=True)
        a0 = e.get_posterior(0)          # (read again, no kernel in between)
        if any(not np.array_equal(a0[k], e.get_posterior(0)[k]) for k in a0): print("iter %d: two READS of the same posterior differ" % it, flush=True)
        e.nms(); e.cluster_fuse(); tot["posterior+nms+cluster"] += cmp_dets("iter %d posterior+nms+cluster" % it)
    print("mismatches:", tot, flush=True)
finally:
    stop = True; t.join()
