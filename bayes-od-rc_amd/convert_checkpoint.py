"""TF-checkpoint -> .npz weight converter.

The reference saves ``tf.train.Checkpoint(step=..., net=model)`` (run_inference.py:78, run_training.py), whose
object-graph keys follow the Python ATTRIBUTE names of the model classes, not the Keras ``name=`` strings:

    net/feature_extractor/conv_1/kernel/.ATTRIBUTES/VARIABLE_VALUE                     (feature_extractor.py:17)
    net/feature_extractor/conv_block_3a/bn_shortcut/moving_mean/.ATTRIBUTES/...        (:234-281)
    net/feature_decoder/c5_reduced/bias/...                                            (feature_decoder.py:16-132)
    net/cls_header/conv_2/kernel/... , net/reg_header/reg_out/... , net/cov_header/cov_out/...
                                                                                       (multitask_headers.py:21-316)

This module maps them to the build's schema -- ``<keras layer name>/<field>`` with fields kernel (HWIO), bias,
gamma, beta, mean, var, the names ``RetinaNetModel.load_weights`` / ``bod_load_weight`` take.  The mapping and
the conversion loop are plain Python over a *reader* object with ``get_variable_to_shape_map()`` and
``get_tensor(key)`` -- the API of ``tf.train.load_checkpoint`` -- so they are tested here with a fake reader
(tests/test_convert_checkpoint.py); only ``main`` imports TensorFlow, on the machine that holds the checkpoint:

    python -m bayes_od_rc_amd.convert_checkpoint <ckpt prefix, e.g. .../checkpoints/ckpt-101> weights.npz
"""
import sys

import numpy as np

SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"
_BN_FIELDS = {"gamma": "gamma", "beta": "beta", "moving_mean": "mean", "moving_variance": "var"}
_CONV_FIELDS = {"kernel": "kernel", "bias": "bias"}
OPTIONAL_PREFIX = "net/reg_header/conv_4/"      # constructed, never called: no variables in a real checkpoint
_STAGE_BLOCKS = {2: "abc", 3: "abcd", 4: "abcdef", 5: "abc"}          # ResNet-50 (feature_extractor.py:36-100)


def attribute_to_layer_map():
    """{object-graph prefix under 'net/': (keras layer name, 'conv' | 'bn')}"""
    m = {"feature_extractor/conv_1": ("conv1", "conv"), "feature_extractor/bn_1": ("bn_conv1", "bn")}
    for stage, blocks in _STAGE_BLOCKS.items():
        for blk in blocks:
            attr = "feature_extractor/%s_%d%s" % ("conv_block" if blk == "a" else "identity_block", stage, blk)
            tag = "%d%s_branch" % (stage, blk)
            for i, part in enumerate(("2a", "2b", "2c"), start=1):
                m["%s/conv_%d" % (attr, i)] = ("res" + tag + part, "conv")
                m["%s/bn_%d" % (attr, i)] = ("bn" + tag + part, "bn")
            if blk == "a":
                m[attr + "/shortcut"] = ("res" + tag + "1", "conv")
                m[attr + "/bn_shortcut"] = ("bn" + tag + "1", "bn")
    for attr, name in (("c5_reduced", "C5_reduced"), ("p5", "P5"), ("p6", "P6"), ("p7", "P7"),
                       ("c4_reduced", "C4_reduced"), ("p4", "P4"), ("c3_reduced", "C3_reduced"), ("p3", "P3")):
        m["feature_decoder/" + attr] = (name, "conv")
    for header, base, out in (("cls_header", "pyramid_classification", "cls_out"),
                              ("reg_header", "pyramid_regression", "reg_out"),
                              ("cov_header", "pyramid_cov", "cov_out")):
        for i in range(4):
            m["%s/conv_%d" % (header, i + 1)] = ("%s_%d" % (base, i), "conv")
        m["%s/%s" % (header, out)] = (base, "conv")
    return m


def checkpoint_key_map():
    """{full checkpoint key: 'layer/field'} for every variable the inference path reads."""
    out = {}
    for prefix, (layer, kind) in attribute_to_layer_map().items():
        for var, field in (_CONV_FIELDS if kind == "conv" else _BN_FIELDS).items():
            out["net/%s/%s%s" % (prefix, var, SUFFIX)] = "%s/%s" % (layer, field)
    return out


def convert(reader, require_all=True):
    """reader -> {'layer': {'field': ndarray}}.  Optimizer slots, the step counter and save counters are skipped;
    a model variable the map does not know raises (the architecture differs from ResNet-50 RetinaNet)."""
    keymap = checkpoint_key_map()
    present = reader.get_variable_to_shape_map()
    weights, unknown = {}, []
    for key in sorted(present):
        if key in keymap:
            layer, field = keymap[key].split("/")
            weights.setdefault(layer, {})[field] = np.asarray(reader.get_tensor(key), dtype=np.float32)
        elif key.startswith("net/") and key.endswith(SUFFIX) and "/.OPTIMIZER_SLOT/" not in key:
            unknown.append(key)
    if unknown:
        raise ValueError("checkpoint holds model variables this converter does not map (not the ResNet-50 "
                         "RetinaNet of retinanet_model.py?): %s" % unknown[:5])
    missing = sorted(set(keymap) - set(present))
    # RegHeader constructs conv_4 but never calls it (multitask_headers.py:181-194 vs :209-230), so Keras never builds
    # it and a real checkpoint holds no variables for it; the engine does not read pyramid_regression_3 either.
    missing = [k for k in missing if not k.startswith(OPTIONAL_PREFIX)]
    # cov_header exists only for 'regression_covar' / 'regression_var' models (retinanet_model.py:53-62): absent as a whole is fine
    if not any(k.startswith("net/cov_header/") for k in present):
        missing = [k for k in missing if not k.startswith("net/cov_header/")]
    if require_all and missing:
        raise ValueError("checkpoint lacks %d expected variables, e.g. %s" % (len(missing), missing[:5]))
    return weights


def save_npz(weights, path):
    np.savez(path, **{"%s/%s" % (l, f): a for l, e in weights.items() for f, a in e.items()})


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 2:
        raise SystemExit(__doc__)
    try:
        import tensorflow as tf
    except ImportError:
        raise SystemExit("convert_checkpoint needs TensorFlow to READ the checkpoint; run it where the checkpoint was "
                         "written and copy the .npz over (this image has no TensorFlow)")
    weights = convert(tf.train.load_checkpoint(argv[0]))
    save_npz(weights, argv[1])
    print("wrote %d layers to %s" % (len(weights), argv[1]))


if __name__ == "__main__":
    main()
