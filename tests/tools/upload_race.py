#!/usr/bin/env python3
"""The asynchronous uint8 upload (copy + pre-processing kernel on the copy stream) while the SAME handle's forward keeps the main stream
busy: are the pre-processed frames still bit-identical to an upload without company?  (development probe)"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bayes_od_rc_amd import synthetic
from bayes_od_rc_amd.engine import Engine, make_config
from bayes_od_rc_amd.distributed import DeviceArray
hw, n, batch = (512, 512), 2, 64
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conftest import ANCHOR_CFG
from bayes_od_rc_amd.anchor_generator import FpnAnchorGenerator
BAYES_CFG = {"ranking_method": "score", "dirichlet_prior": {"type": "non_informative"}, "gaussian_prior": {"type": "isotropic", "isotropic_variance": 100000.0}}
NMS_CFG = {"max_output_size": 100, "iou_threshold": 0.5, "soft_nms_sigma": 0.5}
e = Engine(make_config(hw, batch=batch, mc_samples=n, bayes_od_config=BAYES_CFG, nms_config=NMS_CFG, use_full_covar=True)); e.load_weights(synthetic.make_weights(cls_fg_bias=-1.0))
e.set_anchors(FpnAnchorGenerator(ANCHOR_CFG).generate_all((hw[0], hw[1], 3)))
rng = np.random.default_rng(0)
u8 = [np.ascontiguousarray(rng.integers(0, 256, (batch, hw[0], hw[1], 3), dtype=np.uint8)) for _ in range(2)]
pin = [torch.from_numpy(a).pin_memory() for a in u8]
def dev_images(buf):
    ptr = e.lib.bod_device_images_buffer(e.h, buf)
    return torch.as_tensor(DeviceArray(ptr, (batch, hw[0], hw[1], 3), "<f4"), device="cuda").clone()
ref = []
for k in range(2):
    e.upload_frames_u8_async(pin[k].numpy(), k); e.synchronize(); ref.append(dev_images(k))
e.forward(None, image_buffer=0); e.synchronize()
for company in (False, True):
    bad = 0
    for it in range(60):
        k = it & 1
        slot = e.infer_async(None, image_buffer=k ^ 1) if company else None      # enqueued on the main stream, returns at once
        e.upload_frames_u8_async(pin[k].numpy(), k)           # copy stream, beside the forward
        if slot is not None: e.collect(slot)
        e.synchronize()
        if not torch.equal(dev_images(k), ref[k]): bad += 1
    print("upload beside the forward: %s -> %d of 60 uploads differ from the upload alone" % (company, bad), flush=True)
