"""Known answers of INDEPENDENT implementations for the TensorFlow ops the oracle restates (round-4 review, item 7).

TensorFlow itself is absent and not installable here (the TF half of the oracle stays "parity unpinned" for TF's own
op semantics).  What CAN be pinned without it:
  * the soft-NMS op against the test vector TensorFlow publishes for it (tensorflow/python/ops/image_ops_test.py,
    NonMaxSuppressionWithScoresTest: six boxes, sigma 0.5 -> indices [3, 0, 1, 5, 4, 2], scores 0.95 / 0.9 / 0.384 / 0.3 / 0.256 /
    0.197), the call inference_utils.py:207-212 makes;
  * SAME / VALID convolution placement, stride-2 SAME on odd and even sizes, ZeroPadding2D((1, 2)) + 3x3 s2 max-pool, BatchNorm in
    inference mode and half-pixel nearest up-sampling (feature_extractor.py:31-33,104-139; feature_decoder.py:149-167) against
    PyTorch's own implementations (torch.nn.functional: conv2d with explicit asymmetric padding computed by the published TF rule
    pad_total = max((ceil(n / s) - 1) * s + k - n, 0), pad_before = pad_total // 2; max_pool2d; batch_norm; interpolate(mode=
    'nearest-exact') = the half-pixel rule TF2's resize(NEAREST) uses) on odd sizes.
These replace the builder's hand derivations by independent code; they do not lift the structural cap."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import network, nms


def test_soft_nms_reproduces_tensorflows_published_unit_test_vector():
    boxes = np.array([[0, 0, 1, 1], [0, 0.1, 1, 1.1], [0, -0.1, 1, 0.9], [0, 10, 1, 11], [0, 10.1, 1, 11.1], [0, 100, 1, 101]], np.float32)
    scores = np.array([0.9, 0.75, 0.6, 0.95, 0.5, 0.3], np.float32)
    for variant in ("A", "B"):          # iou_threshold = 1.0: the hard branch of variant A never fires, both published weights agree
        idx, sc = nms.soft_nms(boxes, scores, max_output_size=6, iou_threshold=1.0, soft_nms_sigma=0.5, score_threshold=0.0, variant=variant)
        assert list(idx) == [3, 0, 1, 5, 4, 2], variant
        np.testing.assert_allclose(sc, [0.95, 0.9, 0.384, 0.3, 0.256, 0.197], atol=1e-2)
    # the same vector with the scores TF's test computes exactly: s_j * exp(-iou^2 / (2 * 0.5)) chains
    iou01 = 0.9 / 1.1
    np.testing.assert_allclose(sc[2], 0.75 * np.exp(-iou01 ** 2), rtol=1e-5)            # box 1 decayed by box 0
    np.testing.assert_allclose(sc[4], 0.5 * np.exp(-iou01 ** 2), rtol=1e-5)             # box 4 decayed by box 3
    iou02, iou12 = 0.9 / 1.1, 0.8 / 1.2
    np.testing.assert_allclose(sc[5], 0.6 * np.exp(-iou02 ** 2) * np.exp(-iou12 ** 2), rtol=1e-5)   # box 2 by boxes 0 and 1


def test_hard_nms_limit_of_the_op():
    """sigma = 0: the op degenerates to classic greedy NMS (TF's NonMaxSuppressionTest: boxes above, iou 0.5 -> [3, 0, 5])."""
    boxes = np.array([[0, 0, 1, 1], [0, 0.1, 1, 1.1], [0, -0.1, 1, 0.9], [0, 10, 1, 11], [0, 10.1, 1, 11.1], [0, 100, 1, 101]], np.float32)
    scores = np.array([0.9, 0.75, 0.6, 0.95, 0.5, 0.3], np.float32)
    idx, sc = nms.soft_nms(boxes, scores, max_output_size=3, iou_threshold=0.5, soft_nms_sigma=0.0)
    assert list(idx) == [3, 0, 5]
    np.testing.assert_allclose(sc, [0.95, 0.9, 0.3], rtol=1e-6)


def _tf_same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


@pytest.mark.parametrize("h,w,k,stride,padding", [(7, 5, 3, 2, "same"), (8, 8, 3, 2, "same"), (15, 17, 1, 2, "valid"), (9, 13, 3, 1, "same"),
                                                  (23, 40, 3, 2, "same"), (11, 11, 7, 2, "valid"), (6, 10, 3, 2, "same")])
def test_conv2d_against_torch(h, w, k, stride, padding):
    rng = np.random.default_rng(h * 100 + w + k)
    x = rng.normal(size=(2, h, w, 5))
    wt = rng.normal(size=(k, k, 5, 4))
    b = rng.normal(size=4)
    got = network.conv2d(x, wt, b, stride, padding)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    if padding == "same":
        (pt, pb), (pl, pr) = _tf_same_pad(h, k, stride), _tf_same_pad(w, k, stride)
        xt = F.pad(xt, (pl, pr, pt, pb))
    ref = F.conv2d(xt, torch.from_numpy(wt).permute(3, 2, 0, 1), torch.from_numpy(b), stride=stride).permute(0, 2, 3, 1).numpy()
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("h,w", [(17, 23), (16, 16), (45, 80), (7, 9)])
def test_stem_pool_against_torch(h, w):
    """ZeroPadding2D((1, 2)) -- one row top and bottom, TWO columns left and right (feature_extractor.py:31) -- then 3x3 / s2 VALID max-pool"""
    x = np.random.default_rng(h + w).normal(size=(2, h, w, 3))
    got = network.stem_pool(x)
    xt = F.pad(torch.from_numpy(x).permute(0, 3, 1, 2), (2, 2, 1, 1))
    ref = F.max_pool2d(xt, 3, 2).permute(0, 2, 3, 1).numpy()
    assert got.shape == ref.shape and np.array_equal(got, ref)


def test_batchnorm_eval_against_torch():
    rng = np.random.default_rng(3)
    x = rng.normal(size=(2, 5, 7, 6))
    bn = {"gamma": rng.normal(size=6), "beta": rng.normal(size=6), "mean": rng.normal(size=6), "var": rng.random(6) + 0.1}
    got = network.batchnorm_eval(x, bn)
    t = lambda a: torch.from_numpy(np.asarray(a))
    ref = F.batch_norm(t(x).permute(0, 3, 1, 2), t(bn["mean"]), t(bn["var"]), t(bn["gamma"]), t(bn["beta"]), False, 0.0, network.BN_EPS)
    np.testing.assert_allclose(got, ref.permute(0, 2, 3, 1).numpy(), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("ih,iw,oh,ow", [(23, 40, 45, 80), (12, 20, 23, 40), (6, 10, 12, 20), (8, 27, 16, 53), (4, 14, 8, 27), (3, 5, 6, 10), (5, 5, 5, 5)])
def test_half_pixel_nearest_against_torch_nearest_exact(ih, iw, oh, ow):
    """the FPN's top-down up-sampling (feature_decoder.py:149-167: tf.image.resize(..., 'nearest') to the lateral's size -- TF2's
    half-pixel-centre rule, src = floor((dst + 0.5) * in / out)) incl. the non-integer ratios of the real frames (23 -> 45, 27 -> 53)"""
    x = np.random.default_rng(ih * iw).normal(size=(1, ih, iw, 4))
    got = network.resize_nearest(x, oh, ow)
    ref = F.interpolate(torch.from_numpy(x).permute(0, 3, 1, 2), size=(oh, ow), mode="nearest-exact").permute(0, 2, 3, 1).numpy()
    assert np.array_equal(got, ref)


# ---- round 6: rules and examples PUBLISHED IN THE LIBRARIES' OWN DOCUMENTATION (TensorFlow / Keras / TensorFlow Probability API pages),
# restated here verbatim and checked against the oracle.  They pin what a docstring pins -- a rule and one example -- not an execution.

def _tfp_fill_triangular(x, upper=False):
    """tfp.math.fill_triangular as its API page gives it: for the lower triangle, concatenate x[n:] with reverse(x), reshape to n x n,
    keep the lower band (upper: concatenate x with reverse(x[n:]), keep the upper band)."""
    x = np.asarray(x)
    m = x.shape[-1]
    n = int(np.sqrt(0.25 + 2.0 * m) - 0.5)
    assert n * (n + 1) // 2 == m
    cat = np.concatenate([x, x[n:][::-1]]) if upper else np.concatenate([x[n:], x[::-1]])
    mat = cat.reshape(n, n)
    return np.triu(mat) if upper else np.tril(mat)


def test_fill_triangular_documented_example_and_the_oracles_4x4():
    """The API page's example -- fill_triangular([1, 2, 3, 4, 5, 6]) = [[4, 0, 0], [6, 5, 0], [3, 2, 1]], upper=True: [[1, 2, 3], [0, 5, 6],
    [0, 0, 4]] -- reproduces with the documented algorithm; the same algorithm at n = 4 is oracle/network.py's fill_triangular_4
    (retinanet_model.py:110,144) for random vectors, and the host mirror's."""
    assert _tfp_fill_triangular([1, 2, 3, 4, 5, 6]).tolist() == [[4, 0, 0], [6, 5, 0], [3, 2, 1]]
    assert _tfp_fill_triangular([1, 2, 3, 4, 5, 6], upper=True).tolist() == [[1, 2, 3], [0, 5, 6], [0, 0, 4]]
    from bayes_od_rc_amd.model import fill_triangular_4 as host_fill
    rng = np.random.default_rng(0)
    for _ in range(5):
        x = rng.normal(0, 1, 10)
        want = _tfp_fill_triangular(x)
        assert np.array_equal(network.fill_triangular_4(x[None])[0], want) and np.array_equal(np.asarray(host_fill(x[None]))[0], want)


@pytest.mark.parametrize("rate,kept_value", [(0.5, 2.0), (0.8, 5.0), (0.3, 1.0 / 0.7)])
def test_dropout_scaling_rule_of_tf_nn_dropout(rate, kept_value):
    """tf.nn.dropout's API page: "With probability `rate` elements of x are set to 0.  The remaining elements are scaled up by
    1.0 / (1 - rate), so that the expected value is preserved", with the examples rate 0.5 -> kept ones become 2, rate 0.8 -> 5
    (multitask_headers.py:104-116 calls keras Dropout(0.3)(x, training=True) = this op).  The oracle's head tower on a ones tensor."""
    from oracle import philox

    class Ones(object):             # a numerics object whose convs return ones: the tower's output IS the dropout pattern
        def conv(self, x, name, padding="same", relu=False, store=True):
            return np.ones(x.shape[:3] + (256 if name[-1].isdigit() else 36,), np.float32)

        def store(self, x):
            return x
    P, n = 64, 3
    pyr = [np.ones((1, 8, 8, 256), np.float32)]
    masks = lambda s, lid: philox.dropout_keep_mask(7, 0, s, lid, P, 256, rate)
    seen = {}

    class Spy(Ones):
        def store(self, x):
            seen.setdefault("x", x.copy())
            return x
    network.head_tower(Spy(), pyr, "cls", n, masks, rate, 4)
    vals = np.unique(seen["x"])
    assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - np.float32(kept_value)) <= 1e-6 * kept_value
    dropped = float((seen["x"] == 0).mean())
    assert abs(dropped - rate) < 0.01                                     # "with probability rate"
    assert abs(float(seen["x"].mean()) - 1.0) < 0.02                      # "so that the expected value is preserved"


def test_keras_batchnormalization_documented_defaults_and_inference_formula():
    """tf.keras.layers.BatchNormalization(axis=-1, momentum=0.99, epsilon=0.001, ...) -- the reference passes no epsilon
    (feature_extractor.py:30) -- and the page's inference rule gamma * (batch - moving_mean) / sqrt(moving_var + epsilon) + beta."""
    assert network.BN_EPS == 1e-3
    rng = np.random.default_rng(1)
    x = rng.normal(0, 1, (2, 3, 3, 5))
    bn = {"gamma": rng.uniform(0.5, 1.5, 5), "beta": rng.normal(0, 0.1, 5), "mean": rng.normal(0, 0.1, 5), "var": rng.uniform(0.5, 1.5, 5)}
    want = bn["gamma"] * (x - bn["mean"]) / np.sqrt(bn["var"] + 0.001) + bn["beta"]
    assert np.allclose(network.batchnorm_eval(x, bn), want, rtol=1e-12, atol=1e-12)
    # and the folded form the device uses (App. A.3) is the same function
    w = rng.normal(0, 1, (1, 1, 5, 5))
    conv = {"kernel": w, "bias": rng.normal(0, 1, 5)}
    y = network.batchnorm_eval(network.conv2d(x, w, conv["bias"], 1, "valid"), bn)
    wf, bf = network.fold_bn(conv, bn)
    assert np.allclose(network.conv2d(x, wf.astype(np.float64), bf.astype(np.float64), 1, "valid"), y, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n,k,s", [(512, 3, 1), (512, 3, 2), (16, 3, 2), (8, 3, 2), (23, 3, 2), (39, 3, 2), (45, 3, 1), (512, 7, 2), (7, 3, 2), (2, 3, 2)])
def test_same_padding_rule_of_the_tf_nn_guide(n, k, s):
    """tf.nn's "Notes on padding": out = ceil(in / stride); pad_along = max((out - 1) * stride + filter - in, 0); pad_before =
    pad_along // 2; pad_after = pad_along - pad_before (the odd pixel goes to the bottom / right) -- oracle/network.py's _same_pads,
    which every SAME convolution of the FPN and the heads goes through (feature_decoder.py:140-143)."""
    out = -(-n // s)
    pad = max((out - 1) * s + k - n, 0)
    assert tuple(network._same_pads(n, k, s)) == (pad // 2, pad - pad // 2, out)
