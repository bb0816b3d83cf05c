import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from conftest import ANCHOR_CFG
from bayes_od_rc_amd import synthetic
from bayes_od_rc_amd.engine import Engine, make_config
from bayes_od_rc_amd.anchor_generator import FpnAnchorGenerator
from oracle import network, philox
hw=(128,128); batch=2; n=3; seed=1234567890123; first=7
w = synthetic.make_weights()
frames = synthetic.make_frames(batch, hw[0], hw[1], seed=3)
eng = Engine(make_config(hw, batch=batch, mc_samples=n))
eng.load_weights(w)
eng.forward(frames, seed=seed, first_image_id=first)
cls, box, cov = eng.get_raw()
pyr = [eng.get_pyramid(l) for l in range(5)]
rms = lambda x: float(np.sqrt((np.asarray(x, np.float64)**2).mean()))
for b in range(batch):
    km = lambda s, lid: philox.dropout_keep_mask(seed, first+b, s, lid, eng.P, 256, 0.3)
    ref = network.retinanet_forward(w, frames[b][None], n, 8, mode="bf16", keep_masks=km, return_pyramid=True)
    ref64 = network.retinanet_forward(w, frames[b][None], n, 8, mode="literal", dtype=np.float64, keep_masks=km, return_pyramid=True)
    for l in range(5):
        r = ref["_pyramid"][l][0]; g = pyr[l][b]; t = ref64["_pyramid"][l][0]
        print("img", b, "P%d"%(l+3), "rms", rms(r), "hip-vs-emu", rms(g-r)/rms(r), "hip-vs-f64", rms(g-t)/rms(t), "emu-vs-f64", rms(r-t)/rms(t), "exact", float((g==r).mean()))
    for name, got, key in (("cls", cls[b], "anchors_class_predictions"), ("box", box[b], "anchors_box_predictions"), ("cov", cov[b], "_covar_params")):
        r = ref[key]; t = ref64[key]
        print("img", b, name, "rms", rms(r), "hip-vs-emu", rms(got-r)/rms(r), "hip-vs-f64", rms(got-t)/rms(t), "emu-vs-f64", rms(r-t)/rms(t), "maxabs", float(np.abs(got-r).max()))
