"""CPU: TF-checkpoint key mapping (convert_checkpoint.py) exercised with a fake checkpoint reader."""
import numpy as np
import pytest


class FakeReader(object):
    def __init__(self, tensors):
        self.t = tensors

    def get_variable_to_shape_map(self):
        return {k: list(v.shape) for k, v in self.t.items()}

    def get_tensor(self, key):
        return self.t[key]


def _fake_checkpoint(weights, with_unbuilt_reg_conv4=False):
    """What tf.train.Checkpoint(step, net=model) holds for a trained model: every variable of every layer that was
    ever CALLED.  RegHeader.conv_4 (keras name pyramid_regression_3) is constructed but never called
    (multitask_headers.py:181-194 vs :209-230), so Keras never builds it and a real checkpoint has no entry for it."""
    from bayes_od_rc_amd import convert_checkpoint as cc
    inv = {v: k for k, v in cc.checkpoint_key_map().items()}
    t = {inv["%s/%s" % (l, f)]: a for l, e in weights.items() for f, a in e.items() if a is not None}
    if not with_unbuilt_reg_conv4:
        t = {k: v for k, v in t.items() if not k.startswith("net/reg_header/conv_4/")}
    t["step/.ATTRIBUTES/VARIABLE_VALUE"] = np.zeros((), np.int64)
    t["save_counter/.ATTRIBUTES/VARIABLE_VALUE"] = np.zeros((), np.int64)
    t["net/cls_header/conv_1/kernel/.OPTIMIZER_SLOT/optimizer/m/.ATTRIBUTES/VARIABLE_VALUE"] = np.zeros((3, 3, 256, 256), np.float32)
    return t


def test_round_trip_covers_every_layer_of_the_model():
    from bayes_od_rc_amd import convert_checkpoint as cc, synthetic
    weights = synthetic.make_weights()
    got = cc.convert(FakeReader(_fake_checkpoint(weights)))            # require_all=True, what main() uses
    assert sorted(got) == sorted(set(weights) - {"pyramid_regression_3"}) and len(got) == 128
    for layer, entry in weights.items():
        if layer == "pyramid_regression_3":
            continue
        for field, a in entry.items():
            assert np.array_equal(got[layer][field], a), (layer, field)
    # a checkpoint that does hold the never-called layer (someone built it by hand) converts too
    assert len(cc.convert(FakeReader(_fake_checkpoint(weights, with_unbuilt_reg_conv4=True)))) == 129
    keys = cc.checkpoint_key_map()
    assert "net/feature_extractor/conv_block_3a/bn_shortcut/moving_variance/.ATTRIBUTES/VARIABLE_VALUE" in keys
    assert keys["net/feature_extractor/identity_block_4f/conv_3/kernel/.ATTRIBUTES/VARIABLE_VALUE"] == "res4f_branch2c/kernel"
    assert keys["net/feature_decoder/c4_reduced/bias/.ATTRIBUTES/VARIABLE_VALUE"] == "C4_reduced/bias"
    assert keys["net/reg_header/reg_out/kernel/.ATTRIBUTES/VARIABLE_VALUE"] == "pyramid_regression/kernel"
    assert keys["net/cov_header/conv_4/kernel/.ATTRIBUTES/VARIABLE_VALUE"] == "pyramid_cov_3/kernel"


def test_npz_round_trip_and_errors(tmp_path):
    from bayes_od_rc_amd import convert_checkpoint as cc, synthetic
    weights = synthetic.make_weights()
    t = _fake_checkpoint(weights)
    path = str(tmp_path / "w.npz")
    cc.save_npz(cc.convert(FakeReader(t)), path)
    z = np.load(path)
    assert np.array_equal(z["P6/kernel"], weights["P6"]["kernel"])
    assert len(z.files) == sum(len(e) for l, e in weights.items() if l != "pyramid_regression_3")
    assert "pyramid_regression_3/kernel" not in z.files and "pyramid_regression_2/kernel" in z.files
    # a model without the covariance head converts (cov_header absent as a whole)
    no_cov = {k: v for k, v in t.items() if not k.startswith("net/cov_header/")}
    reg_only = cc.convert(FakeReader(no_cov))         # a 'regression'-only model (retinanet_model.py:53-62)
    assert "pyramid_cov" not in reg_only and "pyramid_cov_0" not in reg_only and "pyramid_regression" in reg_only
    # ... but a cov header with a hole in it is an error
    holed = {k: v for k, v in t.items() if not k.startswith("net/cov_header/conv_2/")}
    with pytest.raises(ValueError):
        cc.convert(FakeReader(holed))
    # a missing backbone variable, or an unknown model variable, is an error
    broken = dict(t)
    del broken["net/feature_extractor/conv_1/kernel/.ATTRIBUTES/VARIABLE_VALUE"]
    with pytest.raises(ValueError):
        cc.convert(FakeReader(broken))
    extra = dict(t)
    extra["net/feature_extractor/conv_block_6a/conv_1/kernel/.ATTRIBUTES/VARIABLE_VALUE"] = np.zeros((1, 1, 4, 4), np.float32)
    with pytest.raises(ValueError):
        cc.convert(FakeReader(extra))
    with pytest.raises(SystemExit):
        cc.main(["ckpt-1", path])               # no TensorFlow in this image
