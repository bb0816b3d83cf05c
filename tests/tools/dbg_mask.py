import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bayes_od_rc_amd.engine import stage_conv
from oracle import philox
rng = np.random.default_rng(0)
x = np.abs(rng.normal(1, 0.1, (2, 50, 50, 64))).astype(np.float32) + 0.5
w = np.zeros((1, 1, 64, 256), np.float32); w[0, 0, 0, :] = 1.0
for prec in ("fp32", "bf16"):
    got = stage_conv(x, w, None, padding="same", relu=True, dropout_rate=0.3, seed=17, layer_id=11, image_id=0, precision=prec)
    for s in range(2):
        keep = philox.dropout_keep_mask(17, 0, s, 11, 2500, 256, 0.3).reshape(50, 50, 256)
        mism = (got[s] != 0) != keep
        print(prec, "sample", s, "mask mismatches:", int(mism.sum()), np.argwhere(mism)[:5].tolist())
        if mism.sum():
            y, xx, c = np.argwhere(mism)[0]
            p = y * 50 + xx
            wd = philox.philox4x32_10(p, int(philox.dropout_group8(c)), s | (11 << 16), 0, 17, 0)
            print("  pixel", p, "channel", c, "words", [hex(int(v)) for v in wd], "thr", hex(int(philox.drop_threshold(0.3))))
