#!/usr/bin/env python3
"""Runs the reference's OWN ``bayes_od_inference`` (src/retina_net/experiments/inference_utils.py:13-217, with the box helpers of
src/retina_net/anchor_generator/box_utils.py it calls) under the NumPy stand-in for TensorFlow in tests/tools/tf_numpy_shim.py and
stores inputs + outputs as ``tests/golden/reference_transcription.npz``.  Build container only (``/root/reference`` must exist);
data out, no reference source text.

What the vectors pin: the TRANSCRIPTION of the reference function into oracle/bayes_od.py -- formulas, axes, mixing weights,
branches (Dirichlet / Gaussian priors on and off, full / diagonal aleatoric covariance, no covariance head, KITTI rescale, both
ranking methods).  What they do not pin: TensorFlow op semantics (stand-ins), ``Categorical.sample`` (the oracle's injected
uniforms) and the soft-NMS (oracle/nms.py is called) -- see the stand-in's header.

The model call is replaced by a stub returning seeded prediction tensors (the function's first statement is
``prediction_dict = model(image, train_val_test='testing')``).
"""
import os
import sys

import numpy as np

REF = os.environ.get("BAYESOD_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))

CASES = [
    # name, N, A, C, covar head, use_full_covar, dirichlet, gaussian, ranking, dataset
    ("bdd_full", 10, 64, 8, True, True, "non_informative", "isotropic", "score", "bdd"),
    ("bdd_diag", 10, 64, 8, True, False, "non_informative", "isotropic", "score", "bdd"),
    ("bdd_nocov", 6, 48, 8, False, False, "non_informative", "isotropic", "score", "bdd"),
    # (gaussian prior 'None' cannot be captured: the reference then leaves the likelihood means rank 2 and its own
    #  tf.squeeze(gaussian_posterior_means, axis=2) at :204-205 raises -- under TensorFlow as under NumPy.  The build returns the
    #  likelihood in that configuration, tests/test_bayes_oracle.py::test_posterior_known_answers.)
    ("bdd_nodirichlet", 10, 64, 8, True, True, "None", "isotropic", "score", "bdd"),
    ("bdd_entropy", 10, 64, 8, True, True, "non_informative", "isotropic", "joint_entropy", "bdd"),
    ("kitti_full", 30, 48, 4, True, True, "non_informative", "isotropic", "score", "kitti"),
    ("kitti_entropy", 30, 48, 4, True, True, "non_informative", "isotropic", "joint_entropy", "kitti"),
]


def make_inputs(seed, n, a, c, covar):
    rng = np.random.default_rng(seed)
    f32 = lambda v: np.asarray(v, np.float32).astype(np.float64)          # float32-valued inputs (stored as float32), float64 arithmetic
    cls = rng.normal(0, 2.0, (n, a, c))
    cls[:, : a // 2, : c - 1] += 3.0 * np.eye(c - 1)[rng.integers(0, c - 1, a // 2)][None]      # half the anchors lean to a foreground class
    box = rng.normal(0, 0.3, (n, a, 4))
    anchors = np.concatenate([rng.uniform(40, 300, (a, 2)), rng.uniform(20, 120, (a, 2))], axis=1)
    pred = {"anchors_class_predictions": f32(cls), "anchors_box_predictions": f32(box)}
    if covar:
        raw = rng.normal(0, 0.4, (n, a, 4, 4))
        pred["anchors_box_covar_predictions"] = f32(np.tril(raw))                              # fill_triangular output: lower triangle
    uniforms = f32(rng.uniform(size=(a, 30)))
    return pred, f32(anchors), uniforms


LOSS_CASES = [
    # name, B, A, C, loss_names, loss_weights  (configs/retinanet_bdd_covar.yaml: classification 5.0 / regression_covar 1.0)
    ("loss_huber", 2, 40, 8, ["classification", "regression"], [5.0, 1.0]),
    ("loss_var", 2, 40, 8, ["classification", "regression_var"], [5.0, 1.0]),
    ("loss_covar", 3, 40, 8, ["classification", "regression_covar"], [5.0, 1.0]),
    ("loss_covar_nopos", 1, 24, 4, ["classification", "regression_covar"], [1.0, 2.0]),
]


def make_loss_inputs(seed, b, a, c, with_pos=True):
    rng = np.random.default_rng(seed)
    f32 = lambda v: np.asarray(v, np.float32).astype(np.float64)
    anchors = np.concatenate([rng.uniform(40, 300, (a, 2)), rng.uniform(20, 120, (a, 2))], axis=1)
    pos = (rng.uniform(size=(b, a)) < (0.2 if with_pos else 0.0)).astype(np.float64)
    neg = ((rng.uniform(size=(b, a)) < 0.5) & (pos == 0)).astype(np.float64)
    cls_t = np.zeros((b, a, c))
    cls_t[..., c - 1] = 1.0
    fg = rng.integers(0, c - 1, (b, a))
    for i in range(b):
        idx = np.nonzero(pos[i])[0]
        cls_t[i, idx, c - 1] = 0.0
        cls_t[i, idx, fg[i, idx]] = 1.0
    sample = {"anchors": f32(anchors)[None], "positive_anchors_mask": pos, "negative_anchors_mask": neg,
              "anchors_class_targets": cls_t, "anchors_box_targets": f32(rng.normal(0, 0.8, (b, a, 4)))}
    pred = {"anchors_class_predictions": f32(rng.normal(0, 2.0, (b, a, c))), "anchors_box_predictions": f32(rng.normal(0, 0.8, (b, a, 4))),
            "anchors_box_covar_predictions": f32(np.tril(rng.normal(0, 0.5, (b, a, 4, 4))))}
    return sample, pred


FORWARD_CASES = [
    # name, (H, W), MC samples, seed  -- one frame each (tf.tile over a batch of one = the oracle's per-frame repeat)
    ("fwd_64_n3", (64, 64), 3, 11),
    ("fwd_96x128_n2", (96, 128), 2, 12),
    ("fwd_64_n1", (64, 64), 1, 13),            # mc_dropout_samples = 1: dropout off (retinanet_model.py:74-77)
]


def forward_inputs(hw, seed):
    """Seeded weights and one normalised frame (the product's synthetic helpers: deterministic, so only outputs are stored)."""
    sys.path.insert(0, ROOT)
    from bayes_od_rc_amd import synthetic
    return synthetic.make_weights(cls_fg_bias=-2.0), synthetic.make_frames(1, hw[0], hw[1], seed=seed).astype(np.float64)


def forward_masks(hw, n, dropout_seed=5, image_id=0):
    """The oracle's Philox keep masks as the (sample, layer id) -> [P, 256] callback oracle/network.py takes."""
    from oracle import philox
    sizes = []
    h, w = hw
    # pyramid level sizes p3..p7 for an H x W frame (SAME stride-2 chain from the stride-8 map)
    lh, lw = -(-h // 8), -(-w // 8)
    for _ in range(5):
        sizes.append(lh * lw)
        lh, lw = -(-lh // 2), -(-lw // 2)
    ptotal = int(sum(sizes))
    cache = {}

    def km(sample, lid):
        if (sample, lid) not in cache:
            cache[(sample, lid)] = philox.dropout_keep_mask(dropout_seed, image_id, sample, lid, ptotal, 256, 0.3)
        return cache[(sample, lid)]
    return km, sizes


def run_reference_training_forward(ref_model_module, yaml_model_config, weights, frames, hw, seed=3, first_image_id=10):
    """RetinaNetModel(...)(frames, 'training') (retinanet_model.py:113-147): dropout on, one pass over the un-tiled batch; the masks are
    the ones oracle/torch_train.py uses (sample 0 of image first_image_id + i)."""
    import copy
    import tf_numpy_shim
    from oracle import philox
    from oracle.network import HEAD_ID
    cfg = copy.deepcopy(yaml_model_config)
    tf_numpy_shim.set_weights(weights)
    model = ref_model_module.RetinaNetModel(cfg)
    _, sizes = forward_masks(hw, 1)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    ptotal = int(offs[-1])
    b = frames.shape[0]
    for head, hdr in (("cls", model.cls_header), ("reg", model.reg_header), ("cov", model.cov_header)):
        for attr, obj in vars(hdr).items():
            if attr.startswith("drop_"):
                obj.tag = (head, int(attr.split("_")[1]) - 1)

    def hook(layer, call_index, x):
        head, k = layer.tag
        lid = HEAD_ID[head] * 4 + k
        nn, h, w, c = x.shape
        assert nn == b and h * w == sizes[call_index]
        return np.stack([philox.dropout_keep_mask(seed, first_image_id + i, 0, lid, ptotal, 256, 0.3)[offs[call_index]:offs[call_index + 1]]
                         for i in range(b)]).reshape(b, h, w, c)
    tf_numpy_shim.set_dropout_hook(hook)
    out = model(frames, train_val_test="training")
    tf_numpy_shim.set_weights(None)
    return out


def run_reference_forward(ref_model_module, yaml_model_config, weights, frame, n, hw):
    """RetinaNetModel(model_config)(frame, 'testing') from the reference's source, layers standing in (tf_numpy_shim)."""
    import copy
    import tf_numpy_shim
    from oracle.network import HEAD_ID
    cfg = copy.deepcopy(yaml_model_config)
    cfg["mc_dropout_samples"] = n
    tf_numpy_shim.set_weights(weights)
    model = ref_model_module.RetinaNetModel(cfg)
    km, sizes = forward_masks(hw, n)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    for head, hdr in (("cls", model.cls_header), ("reg", model.reg_header), ("cov", model.cov_header)):
        for attr, obj in vars(hdr).items():
            if attr.startswith("drop_"):
                obj.tag = (head, int(attr.split("_")[1]) - 1)

    def hook(layer, call_index, x):
        head, k = layer.tag
        lid = HEAD_ID[head] * 4 + k
        nn, h, w, c = x.shape
        assert nn == n and h * w == sizes[call_index], (x.shape, call_index, sizes)
        return np.stack([km(s, lid)[offs[call_index]:offs[call_index + 1]] for s in range(n)]).reshape(n, h, w, c)
    tf_numpy_shim.set_dropout_hook(hook)
    out = model(frame, train_val_test="testing")
    tf_numpy_shim.set_weights(None)
    return out, km


def main():
    import tf_numpy_shim
    tf_numpy_shim.install()
    sys.path.insert(0, REF)
    from src.retina_net.experiments import inference_utils as ref
    import src.core.constants as constants

    out = {}
    for ci, (name, n, a, c, covar, full, dirich, gauss, ranking, dataset) in enumerate(CASES):
        pred, anchors, uniforms = make_inputs(100 + ci, n, a, c, covar)
        cfg = {"ranking_method": ranking, "dirichlet_prior": {"type": dirich},
               "gaussian_prior": {"type": gauss, "isotropic_variance": 100000.0}}
        nms_cfg = {"max_output_size": 100, "iou_threshold": 0.5, "soft_nms_sigma": 0.5}
        net_hw = (384, 1248) if dataset == "kitti" else (512, 512)
        sample = {constants.IMAGE_NORMALIZED_KEY: np.zeros((1, net_hw[0], net_hw[1], 3), np.float32),
                  constants.ANCHORS_KEY: anchors[None],
                  constants.ORIGINAL_IM_SIZE_KEY: np.asarray([[375, 1242, 3]], np.float64)}
        tf_numpy_shim.set_uniforms(uniforms)
        model = lambda image, train_val_test=None, p=pred: dict(p)
        counts, means, covs, nms_idx, iou = ref.bayes_od_inference(model, sample, cfg, nms_cfg, use_full_covar=full, dataset_name=dataset)
        for k, v in pred.items():
            out["%s.in.%s" % (name, k)] = v.astype(np.float32)
        out[name + ".in.anchors"], out[name + ".in.uniforms"] = anchors.astype(np.float32), uniforms.astype(np.float32)
        out[name + ".out.counts"], out[name + ".out.means"], out[name + ".out.covs"] = np.asarray(counts), np.asarray(means), np.asarray(covs)
        out[name + ".out.iou"], out[name + ".out.nms"] = np.asarray(iou), np.asarray(nms_idx)
        print(name, "kept", np.asarray(counts).shape[0], "of", a, "nms", len(nms_idx))
    tf_numpy_shim.set_uniforms(None)
    # ---- RetinaNetModel.get_loss (retinanet_model.py:151-328) with SoftmaxFocalLoss (src/core/losses.py:30-61): the method is
    # called on a bare object carrying the four attributes its __init__ sets (:24-35), no network is built
    from src.retina_net.models import retinanet_model as rm
    import types
    keras = sys.modules["tensorflow"].keras
    for ci, (name, b, a, c, names, weights) in enumerate(LOSS_CASES):
        sample, pred = make_loss_inputs(200 + ci, b, a, c, with_pos="nopos" not in name)
        holder = types.SimpleNamespace(focal_loss=rm.SoftmaxFocalLoss(gamma=2.0, label_smoothing_epsilon=0.001, reduction=keras.losses.Reduction.NONE),
                                       huber_loss=keras.losses.Huber(reduction=keras.losses.Reduction.NONE, name="huber_loss"),
                                       loss_names=names, loss_weights=weights)
        total, parts = rm.RetinaNetModel.get_loss(holder, sample, pred)
        for k, v in list(sample.items()) + list(pred.items()):
            out["%s.in.%s" % (name, k)] = np.asarray(v, np.float32)
        out[name + ".out.total"] = np.float64(total)
        for k, v in parts.items():
            out["%s.out.%s" % (name, k)] = np.float64(v)
        print(name, float(total), {k: float(v) for k, v in parts.items()})
    # ---- validation_utils.post_process_predictions (src/retina_net/experiments/validation_utils.py:10-77): the deterministic
    # validation path (f4) -- one forward pass, arg-max background filter, soft-NMS on the top score, KITTI rescale
    from src.retina_net.experiments import validation_utils as vu
    for ci, dataset in enumerate(("bdd", "kitti")):
        name = "val_" + dataset
        pred, anchors, _ = make_inputs(300 + ci, 1, 80, 8 if dataset == "bdd" else 4, False)
        net_hw = (384, 1248) if dataset == "kitti" else (512, 512)
        sample = {constants.ANCHORS_KEY: anchors[None], constants.IMAGE_NORMALIZED_KEY: np.zeros((1, net_hw[0], net_hw[1], 3), np.float32),
                  constants.ORIGINAL_IM_SIZE_KEY: np.asarray([[375, 1242, 3]], np.float64)}
        classes, corners = vu.post_process_predictions(sample, pred, dataset_name=dataset)
        out[name + ".in.anchors"] = anchors.astype(np.float32)
        for k, v in pred.items():
            out["%s.in.%s" % (name, k)] = v.astype(np.float32)
        out[name + ".out.classes"], out[name + ".out.corners"] = np.asarray(classes, np.float64), np.asarray(corners, np.float64)
        print(name, np.asarray(classes).shape, np.asarray(corners).shape)
    # ---- RetinaNetModel.__init__ / call('testing') (retinanet_model.py:18-112) with FeatureExtractor, FeatureDecoder and the three
    # headers built and called from the reference's source (~600 lines of wiring); Keras layers stand in (see the stand-in's header)
    import yaml
    with open(os.path.join(REF, "src", "retina_net", "configs", "retinanet_bdd_covar.yaml")) as fp:
        model_cfg = yaml.safe_load(fp)["model_config"]
    model_cfg["header"]["num_classes"] = 7                 # config_utils.setup (:77-87): the 7 BDD categories, 3 scales x 3 aspect ratios
    model_cfg["header"]["anchors_per_location"] = 9
    import src.core.constants as constants
    for name, hw, n, seed in FORWARD_CASES:
        weights, frame = forward_inputs(hw, seed)
        pred, _ = run_reference_forward(rm, model_cfg, weights, frame, n, hw)
        for k in (constants.ANCHORS_CLASS_PREDICTIONS_KEY, constants.ANCHORS_BOX_PREDICTIONS_KEY, constants.ANCHORS_COVAR_PREDICTIONS_KEY):
            out["%s.out.%s" % (name, k)] = np.asarray(pred[k], np.float32)           # (wiring errors are O(1): float32 storage is plenty)
        print(name, {k: np.asarray(v).shape for k, v in pred.items()})
    # the training-mode call on a batch of two frames (what oracle/torch_train.py, the training step's oracle, restates in PyTorch)
    sys.path.insert(0, ROOT)
    from bayes_od_rc_amd import synthetic
    weights = synthetic.make_weights(cls_fg_bias=-2.0)
    frames = synthetic.make_frames(2, 64, 64, seed=5).astype(np.float64)
    pred = run_reference_training_forward(rm, model_cfg, weights, frames, (64, 64))
    for k in (constants.ANCHORS_CLASS_PREDICTIONS_KEY, constants.ANCHORS_BOX_PREDICTIONS_KEY, constants.ANCHORS_COVAR_PREDICTIONS_KEY):
        out["train_fwd.out.%s" % k] = np.asarray(pred[k], np.float32)
    print("train_fwd", {k: np.asarray(v).shape for k, v in pred.items()})
    np.savez_compressed(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "reference_transcription.npz"), **out)


if __name__ == "__main__":
    main()
