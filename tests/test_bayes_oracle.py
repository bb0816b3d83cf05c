"""Known answers and properties of the Bayesian stages of the oracle."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from conftest import BAYES_CFG
from oracle import bayes_od, nms, network, geometry


def _pred(rng, n, a, c=8):
    cls = rng.normal(0, 1, (n, a, c))
    cls[..., -1] += 1.0                      # background wins for a good share of the anchors
    return {"anchors_class_predictions": cls,
            "anchors_box_predictions": rng.normal(0, 0.4, (n, a, 4)),
            "anchors_box_covar_predictions": network.fill_triangular_4(rng.normal(0, 0.3, (n, a, 10)))}


def _anchors(a):
    rng = np.random.default_rng(99)
    return np.concatenate([rng.uniform(20, 400, (a, 2)), rng.uniform(16, 128, (a, 2))], 1)


def test_sample_counts_rows_sum_to_draws_and_follow_probs():
    rng = np.random.default_rng(0)
    p = rng.dirichlet(np.ones(8), size=2000)
    u = rng.random((2000, 30)).astype(np.float32)
    counts = bayes_od.sample_counts(p, u)
    assert np.all(counts.sum(1) == 30)
    assert np.abs(counts.mean(0) / 30 - p.mean(0)).max() < 0.01
    one_hot = np.eye(8)[[3]]
    assert np.array_equal(bayes_od.sample_counts(one_hot, u[:1]), 30 * one_hot)
    # u -> 1 selects the last class with non-zero mass, never out of range
    assert bayes_od.sample_counts(np.full((1, 8), 0.125), np.full((1, 30), 1 - 2 ** -24, np.float32))[0, 7] == 30


def test_mean_covariance_is_two_pass_unbiased():
    rng = np.random.default_rng(1)
    x = rng.normal(size=(7, 5, 4))
    mu, cov = bayes_od.mean_covariance(x)
    for m in range(5):
        assert np.allclose(cov[m], np.cov(x[:, m, :].T, ddof=1))
        assert np.allclose(mu[m], x[:, m].mean(0))


def test_n1_gives_nan_covariance():
    """N=1 divides by zero."""
    with np.errstate(all="ignore"):
        _, cov = bayes_od.mean_covariance(np.ones((1, 3, 4)))
    assert np.all(np.isnan(cov))


def test_unit_lower_inverse_and_aleatoric():
    rng = np.random.default_rng(2)
    raw = network.fill_triangular_4(rng.normal(0, 0.5, (6, 10)))
    inv = bayes_od.unit_lower_inverse(raw)
    unit = raw.copy()
    for i in range(4):
        unit[:, i, i] = 1.0
    assert np.allclose(inv, np.linalg.inv(unit))
    full = bayes_od.aleatoric_covariance(raw, True)
    d = np.exp(np.diagonal(raw, axis1=1, axis2=2))
    for m in range(6):
        assert np.allclose(full[m], inv[m] @ np.diag(d[m]) @ inv[m].T)
        assert np.all(np.linalg.eigvalsh(full[m]) > 0)
    diag = bayes_od.aleatoric_covariance(raw, False)
    assert np.allclose(diag[0], np.diag(d[0]))


def test_posterior_known_answers():
    rng = np.random.default_rng(3)
    n, a = 6, 400
    pred, anchors = _pred(rng, n, a), _anchors(a)
    u = rng.random((a, 30)).astype(np.float32)
    out = bayes_od.bayes_od_posterior(pred, anchors, u, BAYES_CFG, use_full_covar=True, return_debug=True)
    m = int(out["keep"].sum())
    assert 0 < m < a
    # Dirichlet prior adds exactly 1/8 to every count; rows then sum to 31
    assert np.allclose(out["counts"] - out["samples"][out["keep"]], 0.125)
    assert np.allclose(out["counts"].sum(1), 31.0)
    assert np.allclose(out["score"].sum(1), 1.0)
    assert np.array_equal(np.argmax(out["samples"], 1) != 7, out["keep"])
    # mixing weights (10 * aleatoric + 1 * epistemic) / 11
    assert np.allclose(out["cov_lik"], (10 * out["cov_al"] + out["cov_epi"]) / 11)
    # weak prior (1e5 I): posterior ~ likelihood, shrunk towards the anchor by ~ cov/1e5
    prec = np.linalg.inv(out["cov_lik"])
    assert np.allclose(out["covs"], np.linalg.inv(prec + np.eye(4) / 1e5), rtol=1e-9)
    rel = np.linalg.norm(out["covs"] - out["cov_lik"], axis=(1, 2)) / np.linalg.norm(out["cov_lik"], axis=(1, 2))
    assert rel.max() < 1e-2
    shift = out["means"][:, :, 0] - out["mu"]
    expect = (out["covs"] / 1e5) @ (anchors[out["keep"]] - out["mu"])[:, :, None]
    assert np.allclose(shift, expect[:, :, 0], rtol=1e-6, atol=1e-9)
    assert np.allclose(out["ranking"], out["score"].max(1))
    assert np.allclose(out["corners"], geometry.vuhw_to_vuvu(out["means"][:, :, 0]))
    # no priors: counts are the raw samples and the posterior is the likelihood
    cfg = {"ranking_method": "score", "dirichlet_prior": {"type": "None"}, "gaussian_prior": {"type": "None"}}
    o2 = bayes_od.bayes_od_posterior(pred, anchors, u, cfg, use_full_covar=True, return_debug=True)
    assert np.array_equal(o2["counts"], o2["samples"][o2["keep"]])
    assert np.allclose(o2["covs"], o2["cov_lik"]) and np.allclose(o2["means"][:, :, 0], o2["mu"])


def test_kitti_rescale():
    rng = np.random.default_rng(4)
    pred, anchors = _pred(rng, 4, 50), _anchors(50)
    u = rng.random((50, 30)).astype(np.float32)
    a = bayes_od.bayes_od_posterior(pred, anchors, u, BAYES_CFG, True, "bdd")
    b = bayes_od.bayes_od_posterior(pred, anchors, u, BAYES_CFG, True, "kitti", orig_size=(375, 1242, 3),
                                    net_size=(512, 1696, 3))
    s = np.array([375 / 512, 1242 / 1696] * 2)
    assert np.allclose(b["means"][:, :, 0], a["means"][:, :, 0] * s, rtol=1e-6)
    assert np.allclose(b["covs"], a["covs"] * s[:, None] * s[None, :], rtol=1e-6)


def test_joint_entropy_ranking_is_normalised():
    rng = np.random.default_rng(5)
    pred, anchors = _pred(rng, 5, 300), _anchors(300)
    u = rng.random((300, 30)).astype(np.float32)
    cfg = dict(BAYES_CFG, ranking_method="joint_entropy")
    out = bayes_od.bayes_od_posterior(pred, anchors, u, cfg, True)
    assert out["ranking"].min() >= 0 and out["ranking"].max() <= 2.0 + 1e-9


@settings(max_examples=25, deadline=None)
@given(st.integers(2, 12), st.integers(0, 2 ** 31 - 1))
def test_posterior_covariances_are_spd(n, seed):
    rng = np.random.default_rng(seed)
    pred, anchors = _pred(rng, n, 40), _anchors(40)
    u = rng.random((40, 30)).astype(np.float32)
    out = bayes_od.bayes_od_posterior(pred, anchors, u, BAYES_CFG, True)
    c = out["covs"]
    assert np.allclose(c, np.transpose(c, (0, 2, 1)), rtol=1e-8, atol=1e-12)
    if len(c):
        assert np.all(np.linalg.eigvalsh(c) > 0)


# ------------------------------------------------------------------------------------------ NMS
def _boxes(*rows):
    return np.array(rows, np.float32)


def test_nms_disjoint_boxes_come_back_in_score_order():
    b = _boxes([0, 0, 10, 10], [20, 20, 30, 30], [40, 40, 50, 50], [60, 60, 70, 70])
    s = np.array([0.2, 0.9, 0.5, 0.7], np.float32)
    idx, sc = nms.soft_nms(b, s, 100, 0.5, 0.5)
    assert idx.tolist() == [1, 3, 2, 0] and np.array_equal(sc, s[[1, 3, 2, 0]])
    idx, _ = nms.soft_nms(b, s, 2, 0.5, 0.5)
    assert idx.tolist() == [1, 3]


def test_nms_variant_a_hard_suppressed_box_returns_with_zero_score():
    """score_threshold = -inf (the reference's default): an IoU>thr box gets weight 0, is re-queued
    with score 0 and is selected last."""
    b = _boxes([0, 0, 10, 10], [0, 0, 10, 9], [20, 20, 30, 30])
    s = np.array([0.9, 0.8, 0.1], np.float32)
    idx, sc = nms.soft_nms(b, s, 100, 0.5, 0.5, variant="A")
    assert idx.tolist() == [0, 2, 1] and sc[2] == 0.0
    idx_b, sc_b = nms.soft_nms(b, s, 100, 0.5, 0.5, variant="B")
    assert idx_b[0] == 0 and abs(float(sc_b[list(idx_b).index(1)]) - 0.8 * np.exp(-0.81)) < 1e-6   # soft decay only


def test_nms_soft_decay_value_and_ties():
    b = _boxes([0, 0, 10, 10], [0, 5, 10, 15], [0, 5, 10, 15])
    s = np.array([0.9, 0.5, 0.5], np.float32)
    idx, sc = nms.soft_nms(b, s, 100, 0.9, 0.5)
    iou = np.float32(50.0) / np.float32(150.0)
    w = np.float32(np.exp(np.float64(np.float32(-1.0) * (iou * iou))))
    assert idx[0] == 0 and idx[1] == 1                    # tie -> lower index first
    assert sc[1] == np.float32(0.5) * w
    assert nms.soft_nms(np.zeros((0, 4), np.float32), np.zeros((0,), np.float32))[0].shape == (0,)


def test_nms_degenerate_boxes_have_zero_iou():
    b = _boxes([5, 5, 5, 20], [0, 0, 10, 30])
    idx, sc = nms.soft_nms(b, np.array([0.9, 0.8], np.float32), 100, 0.5, 0.5)
    assert idx.tolist() == [0, 1] and sc[1] == np.float32(0.8)
