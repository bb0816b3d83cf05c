"""Synthetic weights and frames of the reference's exact architecture and input convention
(there is no network access for BDD/KITTI or the published checkpoints).

Weights: Keras layer names and HWIO kernels, he-normal convs, non-trivial BN statistics (to
exercise folding).  Frames: seeded uint8-uniform RGB -> float32 -> ImageNet mean subtraction ->
BGR, i.e. what ``BddDatasetHandler.create_sample_dict`` produces
(src/retina_net/datasets/bdd/bdd_dataset_handler.py:128-139, src/core/constants.py:12).
"""
import numpy as np

from .constants import MEANS_DICT

_STAGES = ((2, "abc", 64), (3, "abcd", 128), (4, "abcdef", 256), (5, "abc", 512))
_STAGE4_101 = "abcdefghijklmnopqrstuvw"            # 1 ConvBlock + 22 IdentityBlocks (ResNet-101)

# cls foreground bias: the reference initialises it to -log(99) (multitask_headers.py:79-83),
# which leaves ~0 anchors after the background filter on random weights.  For benchmarking the
# Bayesian stages the foreground bias is calibrated so that 500 <= M <= 1500 at 512x512
# (value found with tests/tools: see DESIGN.md "Synthetic workload").
DEFAULT_CLS_FG_BIAS = -np.log(99.0)


def _he(rng, shape, gain=1.0):
    fan_in = shape[0] * shape[1] * shape[2]
    std = gain * np.sqrt(2.0 / fan_in)
    return np.clip(rng.normal(0.0, std, size=shape), -2 * std, 2 * std).astype(np.float32)


def _bn(rng, c, gamma_lo=0.5, gamma_hi=1.5):
    return {"gamma": rng.uniform(gamma_lo, gamma_hi, c).astype(np.float32),
            "beta": rng.normal(0, 0.1, c).astype(np.float32),
            "mean": rng.normal(0, 0.1, c).astype(np.float32),
            "var": rng.uniform(0.5, 1.5, c).astype(np.float32)}


_WEIGHT_CACHE = {}


def make_weights(num_classes_with_bknd=8, anchors_per_location=9, seed=1000, cls_fg_bias=None,
                 cov_out_std=0.02, backbone_bias=True, depth=50):
    """Returns {layer_name: {...}} for ResNet-50 (or -101: depth=101) + FPN + the three heads.  Deterministic in its arguments; the arrays
    of a parameter set are generated once per process (seconds of host time: the GPU suite asks for the same set a hundred times) and
    handed out READ-ONLY inside fresh dictionaries -- replace an entry to change a weight, never write into an array."""
    key = (num_classes_with_bknd, anchors_per_location, seed, cls_fg_bias, cov_out_std, backbone_bias, depth)
    if key not in _WEIGHT_CACHE:
        if len(_WEIGHT_CACHE) >= 6:
            _WEIGHT_CACHE.pop(next(iter(_WEIGHT_CACHE)))
        w = _make_weights(*key)
        for layer in w.values():
            for arr in layer.values():
                if isinstance(arr, np.ndarray):
                    arr.setflags(write=False)
        _WEIGHT_CACHE[key] = w
    return {name: dict(layer) for name, layer in _WEIGHT_CACHE[key].items()}


def _make_weights(num_classes_with_bknd, anchors_per_location, seed, cls_fg_bias, cov_out_std, backbone_bias, depth):
    w = {}
    idx = [0]

    def rng():
        idx[0] += 1
        return np.random.default_rng(seed + idx[0])

    def conv(name, kh, kw, cin, cout, bias=True, gain=1.0):
        r = rng()
        w[name] = {"kernel": _he(r, (kh, kw, cin, cout), gain),
                   "bias": (r.normal(0, 0.02, cout).astype(np.float32) if bias else None)}

    conv("conv1", 7, 7, 3, 64, backbone_bias, gain=0.05)     # inputs are ~+-128, keep activations O(1)
    w["bn_conv1"] = _bn(rng(), 64)
    cin = 64
    for stage, blocks, f1 in _STAGES:
        if stage == 4 and depth == 101:
            blocks = _STAGE4_101
        for blk in blocks:
            cb, bb = "res%d%s_branch" % (stage, blk), "bn%d%s_branch" % (stage, blk)
            conv(cb + "2a", 1, 1, cin, f1, backbone_bias); w[bb + "2a"] = _bn(rng(), f1)
            conv(cb + "2b", 3, 3, f1, f1, backbone_bias); w[bb + "2b"] = _bn(rng(), f1)
            # the residual branch's last BN keeps the un-normalised random network from growing block after block
            # (23 blocks in the ResNet-101 stage 4: smaller gains there)
            g_lo, g_hi = (0.05, 0.15) if (stage == 4 and depth == 101) else (0.2, 0.4)
            conv(cb + "2c", 1, 1, f1, 4 * f1, backbone_bias); w[bb + "2c"] = _bn(rng(), 4 * f1, g_lo, g_hi)
            if blk == "a":
                conv(cb + "1", 1, 1, cin, 4 * f1, backbone_bias); w[bb + "1"] = _bn(rng(), 4 * f1)
            cin = 4 * f1
    # lateral gains bring the (un-normalised, random-BN) backbone maps down to O(1) pyramid values
    conv("C5_reduced", 1, 1, 2048, 256, gain=0.0165)
    conv("P5", 3, 3, 256, 256, gain=0.7)
    conv("P6", 3, 3, 2048, 256, gain=0.0236)
    conv("P7", 3, 3, 256, 256)
    conv("C4_reduced", 1, 1, 1024, 256, gain=0.03)
    conv("P4", 3, 3, 256, 256, gain=0.7)
    conv("C3_reduced", 1, 1, 512, 256, gain=0.038)
    conv("P3", 3, 3, 256, 256, gain=0.7)
    a, c = anchors_per_location, num_classes_with_bknd
    for prefix in ("pyramid_classification", "pyramid_regression", "pyramid_cov"):
        for i in range(4):          # regression_3 exists but is never called (multitask_headers.py:181-194)
            conv("%s_%d" % (prefix, i), 3, 3, 256, 256)
    conv("pyramid_classification", 1, 1, 256, a * c, gain=0.5)
    fg = DEFAULT_CLS_FG_BIAS if cls_fg_bias is None else cls_fg_bias
    bias = np.zeros(c, np.float32)
    bias[:-1] = fg
    w["pyramid_classification"]["bias"] = np.tile(bias, a).astype(np.float32)
    conv("pyramid_regression", 1, 1, 256, a * 4, gain=0.1)
    r = rng()
    w["pyramid_cov"] = {"kernel": r.normal(0, cov_out_std, (1, 1, 256, a * 10)).astype(np.float32),
                        "bias": np.zeros(a * 10, np.float32)}
    return w


def make_frames(count, height, width, seed=0, normalization="ImageNet"):
    """[count,H,W,3] float32 normalised BGR frames; frame i uses default_rng(seed + i)."""
    means = np.asarray(MEANS_DICT[normalization], dtype=np.float32).reshape(1, 1, 3)
    out = np.empty((count, height, width, 3), np.float32)
    for i in range(count):
        rgb = np.random.default_rng(seed + i).integers(0, 256, size=(height, width, 3), dtype=np.uint8)
        out[i] = (rgb.astype(np.float32) - means)[:, :, ::-1]
    return out


def calibrate_fg_bias(cls_logits, current_fg_bias, target_fraction=0.02, iters=40):
    """Given raw class logits [N,A,C] produced with ``current_fg_bias``, returns the foreground
    bias for which ~target_fraction of anchors have a non-background arg-max of the MC-mean
    softmax (bias is additive on the logits, so no network re-run is needed)."""
    x = np.asarray(cls_logits, dtype=np.float64)

    def frac(delta):
        z = x.copy()
        z[..., :-1] += delta
        z -= z.max(axis=-1, keepdims=True)
        p = np.exp(z)
        p /= p.sum(axis=-1, keepdims=True)
        return float((p.mean(axis=0).argmax(axis=-1) != x.shape[-1] - 1).mean())

    lo, hi = -20.0, 20.0
    for _ in range(iters):
        mid = 0.5 * (lo + hi)
        if frac(mid) < target_fraction:
            lo = mid
        else:
            hi = mid
    return float(current_fg_bias + 0.5 * (lo + hi))
