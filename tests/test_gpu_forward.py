"""GPU parity: RetinaNet forward (backbone + FPN + MC-dropout heads) through the C ABI vs the
oracle's bf16-storage emulation on identical weights, frames and Philox dropout masks."""
import numpy as np
import pytest

from conftest import ANCHOR_CFG, rel_err

pytestmark = pytest.mark.gpu

# parity bar (BASELINE.json north_star): 1e-3 relative.  The denominator floor is 1e-3 of the
# tensor's RMS... see DESIGN.md "Numerics": both sides round activations to bf16 at the same
# points, so remaining differences are fp32 summation order plus rare 1-ulp bf16 flips.
REL_TOL = 1e-3


def _setup(hw, batch, n, seed=42):
    from bayes_od_rc_amd import synthetic
    from bayes_od_rc_amd.engine import Engine, make_config
    from bayes_od_rc_amd.anchor_generator import FpnAnchorGenerator
    w = synthetic.make_weights()
    frames = synthetic.make_frames(batch, hw[0], hw[1], seed=3)
    eng = Engine(make_config(hw, batch=batch, mc_samples=n))
    eng.load_weights(w)
    eng.set_anchors(FpnAnchorGenerator(ANCHOR_CFG).generate_all((hw[0], hw[1], 3)))
    return w, frames, eng


def _oracle(w, frame, n, seed, image_id, P):
    from oracle import network, philox
    km = lambda s, lid: philox.dropout_keep_mask(seed, image_id, s, lid, P, 256, 0.3)
    return network.retinanet_forward(w, frame[None], n, 8, mode="bf16", keep_masks=km, return_pyramid=True)


@pytest.mark.parametrize("hw,batch,n", [((128, 128), 2, 3), ((96, 160), 1, 2)])
def test_forward_matches_oracle(hw, batch, n):
    seed, first = 1234567890123, 7
    w, frames, eng = _setup(hw, batch, n)
    eng.forward(frames, seed=seed, first_image_id=first)
    cls, box, cov = eng.get_raw()
    pyr = [eng.get_pyramid(l) for l in range(5)]
    for b in range(batch):
        ref = _oracle(w, frames[b], n, seed, first + b, eng.P)
        for l in range(5):
            r = ref["_pyramid"][l][0]
            floor = 1e-2 * float(np.sqrt((r ** 2).mean()))
            # pyramid values are bf16 on both sides: allow one bf16 ulp (2^-8) on rare elements
            err = np.abs(pyr[l][b] - r) / (np.abs(r) + floor)
            assert np.quantile(err, 0.999) < REL_TOL, (l, float(np.quantile(err, 0.999)))
            assert err.max() < 2.0 ** -7, (l, float(err.max()))
        for name, got, key in (("cls", cls[b], "anchors_class_predictions"),
                               ("box", box[b], "anchors_box_predictions"),
                               ("cov", cov[b], "_covar_params")):
            r = ref[key]
            rms = float(np.sqrt((r.astype(np.float64) ** 2).mean()))
            e = rel_err(got, r, floor=rms)
            assert e < REL_TOL, (name, b, e)


def test_n1_disables_dropout():
    """mc_dropout_samples == 1 => dropout off (retinanet_model.py:74-77); BASELINE config 2."""
    hw = (128, 128)
    w, frames, eng = _setup(hw, 1, 1)
    eng.forward(frames, seed=5, first_image_id=0)
    cls, box, cov = eng.get_raw()
    from oracle import network
    ref = network.retinanet_forward(w, frames[0][None], 1, 8, mode="bf16")
    for got, key in ((cls[0], "anchors_class_predictions"), (box[0], "anchors_box_predictions"),
                     (cov[0], "_covar_params")):
        r = ref[key]
        rms = float(np.sqrt((r.astype(np.float64) ** 2).mean()))
        assert rel_err(got, r, floor=rms) < REL_TOL
    # seed must not matter without dropout
    eng.forward(frames, seed=6, first_image_id=3)
    cls2, _, _ = eng.get_raw()
    assert np.array_equal(cls, cls2)
