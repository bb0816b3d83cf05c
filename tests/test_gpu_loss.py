"""GPU parity of the loss forward (BASELINE config 5) through the C ABI /
RetinaNetModel.get_loss vs the oracle restatement on identical inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _model(loss_names, loss_weights):
    from bayes_od_rc_amd.model import RetinaNetModel
    cfg = {"output_names": ["classification", "regression", "regression_covar"], "mc_dropout_samples": 10,
           "header": {"dropout_rate": 0.3, "num_classes": 7, "anchors_per_location": 9},
           "losses": {"loss_names": loss_names, "loss_weights": loss_weights, "label_smoothing_epsilon": 0.001}}
    return RetinaNetModel(cfg)


@pytest.mark.parametrize("names,weights", [(["classification", "regression_covar"], [5.0, 1.0]),
                                           (["classification", "regression_var"], [5.0, 1.0]),
                                           (["classification", "regression"], [1.0, 50.0]),
                                           (["regression_covar"], [1.0])])
def test_get_loss_matches_oracle(names, weights):
    from test_losses_oracle import _sample
    from oracle import losses
    from bayes_od_rc_amd import constants
    rng = np.random.default_rng(5)
    sample, pred = _sample(rng, 3, 4911)
    sample32 = {constants.ANCHORS_KEY: sample["anchors"].astype(np.float32)[None],
                constants.POSITIVE_ANCHORS_MASK_KEY: sample["positive_anchors_mask"],
                constants.NEGATIVE_ANCHOR_MASK_KEY: sample["negative_anchors_mask"],
                constants.ANCHORS_CLASS_TARGETS_KEY: sample["anchors_class_targets"].astype(np.float32),
                constants.ANCHORS_BOX_TARGETS_KEY: sample["anchors_box_targets"].astype(np.float32)}
    pred32 = {k: v.astype(np.float32) for k, v in pred.items()}
    total, d = _model(names, weights).get_loss(sample32, pred32)
    # the oracle sees the same float32-rounded inputs, evaluated in float64
    s64 = {"anchors": sample32[constants.ANCHORS_KEY], "positive_anchors_mask": sample["positive_anchors_mask"],
           "negative_anchors_mask": sample["negative_anchors_mask"],
           "anchors_class_targets": sample32[constants.ANCHORS_CLASS_TARGETS_KEY],
           "anchors_box_targets": sample32[constants.ANCHORS_BOX_TARGETS_KEY]}
    ref_total, ref = losses.get_loss(s64, pred32, names, weights)
    assert abs(total - ref_total) <= 1e-3 * abs(ref_total)            # BASELINE.json: 1e-3 relative
    for k, v in ref.items():
        assert abs(d[k] - v) <= 1e-3 * abs(v) + 1e-9, k
    assert set(d) == set(ref)


def test_get_loss_errors():
    with pytest.raises(ValueError):
        _model(["bogus"], [1.0]).get_loss({}, {})
