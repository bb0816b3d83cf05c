"""Known answers for the loss forward."""
import numpy as np
import pytest

from oracle import losses, network


def test_huber_elementwise():
    e = np.array([[[0.5, -0.5, 2.0, -3.0]]])
    assert np.allclose(losses.huber(np.zeros_like(e), e), [[[0.125, 0.125, 1.5, 2.5]]])


def test_focal_loss_values():
    # uniform logits over C=8, foreground target 0: p_t = 1/8, CE with smoothing = log 8 exactly
    x = np.zeros((1, 1, 8))
    y = np.eye(8)[[0]][None]
    l = losses.softmax_focal_loss(y, x)
    assert np.allclose(l, 0.5 * (1 - 1 / 8) ** 2 * np.log(8))
    # confident and right => ~0; background target uses (1 - alpha) = 0.5 as well
    x = np.array([[[10.0, 0, 0, 0, 0, 0, 0, 0]]])
    assert losses.softmax_focal_loss(y, x)[0, 0] < 1e-6
    yb = np.eye(8)[[7]][None]
    assert np.allclose(losses.softmax_focal_loss(yb, np.zeros((1, 1, 8))), 0.5 * (7 / 8) ** 2 * np.log(8))
    # label smoothing: y_s = y(1-e)+e/C
    x = np.array([[[2.0, -1.0, 0.5, 0.0]]])
    y4 = np.eye(4)[[1]][None]
    ls = x - np.log(np.exp(x).sum())
    ys = y4 * 0.999 + 0.001 / 4
    p = np.exp(ls)[0, 0, 1]
    assert np.allclose(losses.softmax_focal_loss(y4, x), 0.5 * (1 - p) ** 2 * -(ys * ls).sum())


def _sample(rng, b, a, c=8):
    anchors = np.concatenate([rng.uniform(20, 200, (a, 2)), rng.uniform(16, 64, (a, 2))], 1)
    pos = rng.random((b, a)) < 0.1
    neg = (~pos) & (rng.random((b, a)) < 0.8)
    cls_t = np.zeros((b, a, c)); cls_t[..., -1] = 1
    fg = rng.integers(0, c - 1, (b, a))
    cls_t[pos] = np.eye(c)[fg[pos]]
    sample = {"anchors": anchors, "positive_anchors_mask": pos, "negative_anchors_mask": neg,
              "anchors_class_targets": cls_t, "anchors_box_targets": rng.normal(0, 0.5, (b, a, 4))}
    pred = {"anchors_class_predictions": rng.normal(0, 1.5, (b, a, c)),
            "anchors_box_predictions": rng.normal(0, 0.5, (b, a, 4)),
            "anchors_box_covar_predictions": network.fill_triangular_4(rng.normal(0, 0.3, (b, a, 10)))}
    return sample, pred


def test_get_loss_composition():
    rng = np.random.default_rng(0)
    sample, pred = _sample(rng, 2, 300)
    total, d = losses.get_loss(sample, pred, ["classification", "regression_covar"], [5.0, 1.0])
    assert np.isclose(total, d["cls_loss"] + d["reg_loss"] + d["covariance_loss"])
    t2, d2 = losses.get_loss(sample, pred, ["classification", "regression_var"], [5.0, 1.0])
    assert np.isclose(d2["covariance_loss"], d["covariance_loss"]) and d2["reg_loss"] < d["reg_loss"]   # no ||L||_F >= 2 factor
    t3, d3 = losses.get_loss(sample, pred, ["regression"], [2.0])
    pos = sample["positive_anchors_mask"]
    h = losses.huber(sample["anchors_box_targets"], pred["anchors_box_predictions"]).mean(2)
    assert np.isclose(t3, 2.0 * (h * pos).sum() / pos.sum())
    # ignored anchors (neither positive nor negative) do not contribute to the focal term
    s2 = dict(sample); s2["negative_anchors_mask"] = np.zeros_like(sample["negative_anchors_mask"])
    _, d4 = losses.get_loss(s2, pred, ["classification"], [1.0])
    l = losses.softmax_focal_loss(sample["anchors_class_targets"], pred["anchors_class_predictions"])
    assert np.isclose(d4["cls_loss"], (l * pos).sum() / pos.sum())
    with pytest.raises(ValueError):
        losses.get_loss(sample, pred, ["bogus"], [1.0])


def test_zero_positives_divides_by_one():
    rng = np.random.default_rng(1)
    sample, pred = _sample(rng, 1, 50)
    sample["positive_anchors_mask"][:] = False
    total, d = losses.get_loss(sample, pred, ["classification", "regression_covar"], [5.0, 1.0])
    assert d["reg_loss"] == 0 and d["covariance_loss"] == 0 and np.isfinite(total)
