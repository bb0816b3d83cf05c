"""Oracle restatement of the training-loss FORWARD of the reference (SURVEY.md row a19, App. A.10;
BASELINE config 5): ``RetinaNetModel.get_loss`` (src/retina_net/models/retinanet_model.py:151-328)
and ``SoftmaxFocalLoss.call`` (src/core/losses.py:30-61).  TF/Keras half => parity unpinned; checked by
hand-derivable known answers (tests/test_losses_oracle.py) and, as a transcription check, against the reference's own
``get_loss`` source run under a NumPy stand-in for TensorFlow / Keras (tests/test_reference_transcription.py)."""
import numpy as np

from . import geometry


def log_softmax(x):
    m = x.max(axis=-1, keepdims=True)
    z = x - m
    return z - np.log(np.exp(z).sum(axis=-1, keepdims=True))


def softmax_focal_loss(target, logits, gamma=2.0, alpha=0.5, label_smoothing=0.001, temperature=1.0):
    """losses.py:30-61.  target/logits [B,A,C] -> [B,A].
    keras CategoricalCrossentropy(from_logits, label_smoothing=e): y*(1-e) + e/C."""
    t = logits.dtype.type
    x = logits / t(temperature)
    ls = log_softmax(x)
    p_t = (np.exp(ls) * target).sum(axis=2)                                   # :43-46
    c = target.shape[2]
    y = target * t(1.0 - label_smoothing) + t(label_smoothing / c)
    ce = -(y * ls).sum(axis=2)                                                # :48-49
    focus = (t(1.0) - p_t) ** t(gamma)                                        # :51-52
    neg = target[:, :, -1]                                                    # :55-59
    alpha_f = t(alpha) * (t(1.0) - neg) + t(1.0 - alpha) * neg
    return alpha_f * focus * ce


def huber(target, pred, delta=1.0):
    """keras Huber(reduction=NONE) used element-wise (retinanet_model.py:215-218; App. A.10)."""
    e = pred - target
    a = np.abs(e)
    t = pred.dtype.type
    return np.where(a <= t(delta), t(0.5) * e * e, t(delta) * a - t(0.5 * delta * delta))


def get_loss(sample, prediction, loss_names, loss_weights, label_smoothing=0.001, dtype=np.float64):
    """retinanet_model.py:151-328.  sample: anchors [A,4] or [1,A,4], positive/negative masks [B,A],
    class targets [B,A,C], box targets [B,A,4].  prediction: cls [B,A,C], box [B,A,4],
    covar [B,A,4,4] (fill_triangular output).  Returns (total, dict)."""
    t = dtype
    anchors = np.asarray(sample["anchors"], dtype=t).reshape(1, -1, 4)
    pos = np.asarray(sample["positive_anchors_mask"], dtype=t)
    neg = np.asarray(sample["negative_anchors_mask"], dtype=t)
    n_pos = pos.sum()
    cls_mask = pos + neg
    tgt_cls = np.asarray(sample["anchors_class_targets"], dtype=t)
    tgt_box = np.asarray(sample["anchors_box_targets"], dtype=t)
    p_cls = np.asarray(prediction["anchors_class_predictions"], dtype=t)
    p_box = np.asarray(prediction["anchors_box_predictions"], dtype=t)
    total = t(0.0)
    out = {}
    for name in loss_names:
        w = t(loss_weights[loss_names.index(name)])
        if name == "classification":                                          # :183-203
            l = softmax_focal_loss(tgt_cls, p_cls, gamma=2.0, label_smoothing=label_smoothing)
            v = (l * cls_mask).sum() / max(n_pos, t(1.0)) * w
            out["cls_loss"] = v
            total = total + v
        elif name == "regression":                                            # :205-226
            l = huber(tgt_box, p_box).mean(axis=2)
            v = (l * pos).sum() / max(n_pos, t(1.0)) * w
            out["reg_loss"] = v
            total = total + v
        elif name in ("regression_var", "regression_covar"):                  # :228-323
            pb = geometry.box_from_anchor_and_target(anchors, p_box)
            tb = geometry.box_from_anchor_and_target(anchors, tgt_box)
            cov = np.asarray(prediction["anchors_box_covar_predictions"], dtype=t)
            log_d = np.diagonal(cov, axis1=-2, axis2=-1)
            compute = (np.exp(-log_d) * huber(tb, pb)).sum(axis=2)
            if name == "regression_covar":
                l_inv = cov.copy()
                for i in range(4):
                    l_inv[..., i, i] = 1.0
                compute = np.sqrt((l_inv ** 2).sum(axis=(-2, -1))) * compute     # Frobenius norm (:289-296)
            reg = t(0.5) * log_d.sum(axis=2)
            denom = max(t(1.0), n_pos)
            v = w * ((compute + reg) * pos).sum() / denom
            out["reg_loss"] = (compute * pos).sum() / denom
            out["covariance_loss"] = (reg * pos).sum() / denom
            total = total + v
        else:
            raise ValueError("Invalid Loss! Not implemented yet.", name)
    return total, out
