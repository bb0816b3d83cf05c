import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, ctypes as C
from bayes_od_rc_amd import synthetic, constants, _lib
from bayes_od_rc_amd.engine import Engine, make_config
from bayes_od_rc_amd.run_training import synthetic_samples
hw, batch = (512, 512), 3
acfg = {'layers': [3, 4, 5, 6, 7], 'aspect_ratios': [[1, 1], [1, 2], [2, 1]], 'scales': [1.0, 1.26, 1.59], 'min_positive_iou': 0.5, 'max_negative_iou': 0.4}
samples = synthetic_samples(batch, hw, acfg, 7)
eng = Engine(make_config(hw, batch=batch, mc_samples=1, training=True))
eng.load_weights(synthetic.make_weights())
eng.set_anchors(np.asarray(samples[0][constants.ANCHORS_KEY], np.float32))
st = lambda k: np.ascontiguousarray(np.stack([s[k] for s in samples]))
img = st(constants.IMAGE_NORMALIZED_KEY).astype(np.float32)
ct = st(constants.ANCHORS_CLASS_TARGETS_KEY).astype(np.float32); bt = st(constants.ANCHORS_BOX_TARGETS_KEY).astype(np.float32)
pm = st(constants.POSITIVE_ANCHORS_MASK_KEY).astype(np.uint8); nm = st(constants.NEGATIVE_ANCHOR_MASK_KEY).astype(np.uint8)
eng.upload_images(img)
u8 = C.POINTER(C.c_uint8)
ptr = eng.lib.bod_device_images(eng.h)
def step(sync):
    out = (C.c_double * 6)() if sync else None
    st_ = eng.lib.bod_train_step(eng.h, ptr, 1, _lib.fptr(ct), _lib.fptr(bt), pm.ctypes.data_as(u8), nm.ctypes.data_as(u8), 1, 0, 3, 0.001, 5.0, 1.0, 1e-6, 1e-3, 1, out)
    assert st_ == 0
for i in range(3): step(True)
eng.synchronize()
t0 = time.perf_counter()
for i in range(10): step(False)
t1 = time.perf_counter()
eng.synchronize()
t2 = time.perf_counter()
print("host enqueue per step %.2f ms; with drain %.2f ms per step" % ((t1 - t0) / 10 * 1e3, (t2 - t0) / 10 * 1e3))
