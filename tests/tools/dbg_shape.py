import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bayes_od_rc_amd import synthetic
from bayes_od_rc_amd.engine import Engine, make_config
from oracle import network, philox
hw=(192,624); n=2; seed=17
w = synthetic.make_weights()
frames = synthetic.make_frames(1, hw[0], hw[1], seed=9)
eng = Engine(make_config(hw, batch=1, mc_samples=n, precision="fp32"))
eng.load_weights(w); eng.forward(frames, seed=seed, first_image_id=0)
cls, box, cov = eng.get_raw()
km = lambda s, lid: philox.dropout_keep_mask(seed, 0, s, lid, eng.P, 256, 0.3)
f64 = network.retinanet_forward(w, frames, n, 8, mode="literal", dtype=np.float64, keep_masks=km, return_pyramid=True)
rms = lambda x: float(np.sqrt((np.asarray(x, np.float64)**2).mean()))
for got, key in ((cls[0], "anchors_class_predictions"), (box[0], "anchors_box_predictions"), (cov[0], "_covar_params")):
    t = f64[key]; e = np.abs(got - t) / (np.abs(t) + rms(t))
    idx = np.unravel_index(np.argmax(e), e.shape)
    print(key, "max", e.max(), "at", idx, "frac>1e-3", (e > 1e-3).mean(), got[idx], t[idx])
    bad = np.argwhere(e > 1e-3)
    if len(bad): print(bad[:5], bad[-5:], np.unique(bad[:, 2]) if bad.shape[1] > 2 else None)
print("levels", eng.levels, "P", eng.P)
eng.forward(frames, seed=seed, first_image_id=0)
cov2 = eng.get_raw()[2]
print("deterministic:", np.array_equal(cov, cov2))
t = f64["_covar_params"]; e = np.abs(cov[0] - t) / (np.abs(t) + rms(t))
bad_pix = np.unique(np.argwhere(e > 1e-3)[:, 1] // 9)
print("bad pixels (per sample-agnostic):", bad_pix, "samples:", np.unique(np.argwhere(e > 1e-3)[:, 0]))
offs = np.cumsum([0] + [h * w for h, w in eng.levels])
for p in bad_pix:
    l = int(np.searchsorted(offs, p, side="right") - 1); q = p - offs[l]; print("pixel", p, "level", l, "y,x", divmod(int(q), eng.levels[l][1]))
eng16 = Engine(make_config(hw, batch=1, mc_samples=n)); eng16.load_weights(w); eng16.forward(frames, seed=seed, first_image_id=0)
c16 = eng16.get_raw()[2][0]
e16 = np.abs(c16 - t) / (np.abs(t) + rms(t))
print("bf16 path: worst pixel", int(np.argmax(e16.max(axis=(0, 2)))) // 9, float(e16.max()), "rms rel", rms(c16 - t) / rms(t))
