"""Probe (GPU): per-tensor distance of a training handle's gradients to the float64 autograd oracle.
    python tests/tools/train_emulation_probe.py <depth> <precision: bf16|fp32> [emulate]
`emulate` compares with the oracle's bf16-storage emulation instead of the literal float64 step."""
import sys
import numpy as np
sys.path.insert(0, "tests")
from test_gpu_train_step import _problem
from bayes_od_rc_amd.engine import Engine, make_config
from oracle import torch_train

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
precision = sys.argv[2] if len(sys.argv) > 2 else "bf16"
emulate = len(sys.argv) > 3
hw, batch = (64, 64), 2
weights, anchors, frames, cls_t, box_t, pos, neg = _problem(hw, batch, depth=depth)
eng = Engine(make_config(hw, batch=batch, mc_samples=1, training=True, backbone_depth=depth, precision=precision))
eng.load_weights(weights)
eng.set_anchors(anchors)
got = eng.train_step(frames, cls_t, box_t, pos, neg, seed=3, first_image_id=10, apply_update=False)
ref, grads, _, _ = torch_train.train_step(weights, frames, cls_t, box_t, anchors, pos, neg, seed=3, first_image_id=10, emulate_bf16=emulate)
print({k: (got[k], ref[k]) for k in ref})
rows = []
for name, g in grads.items():
    layer, kind = name.rsplit("/", 1)
    if layer == "pyramid_regression_3":
        continue
    mine = eng.train_get(layer, kind, g.shape, what="grad").astype(np.float64)
    nr = np.linalg.norm(g)
    if nr < 1e-7 * ref["grad_norm"]:
        continue
    rows.append((float(np.linalg.norm(mine - g) / nr), float(np.abs(mine - g).max() / np.abs(g).max()), name, g.size))
rows.sort()
for r in rows[:3] + rows[len(rows) // 2 - 1:len(rows) // 2 + 1] + rows[-25:]:
    print("%.3e %.3e %s %d" % r)
print("n", len(rows), "median", rows[len(rows) // 2][0], "p90", rows[int(len(rows) * 0.9)][0], "max elementwise", max(r[1] for r in rows))
