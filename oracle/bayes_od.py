"""Oracle restatement of ``bayes_od_inference`` after the network forward
(reference src/retina_net/experiments/inference_utils.py:13-277).

Randomness (``Categorical.sample(30)``, :37-46) is injected as uniforms: class of a
draw = first c with cumsum(p)[c] > u * cumsum(p)[C-1], cumsum sequential in ``dtype``.
"""
import numpy as np

from . import geometry

MIX_ALEATORIC = 10.0      # inference_utils.py:86-87
MIX_EPISTEMIC = 1.0


def softmax(x):
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=-1, keepdims=True)


def sample_counts(mean_probs, uniforms):
    """:37-46 with injected uniforms.  mean_probs [A,C], uniforms [A,D] -> counts [A,C] (rows sum to D)."""
    a, c = mean_probs.shape
    cdf = np.zeros_like(mean_probs)
    acc = np.zeros(a, dtype=mean_probs.dtype)
    for j in range(c):                      # sequential cumsum, same order as the device
        acc = acc + mean_probs[:, j]
        cdf[:, j] = acc
    t = uniforms.astype(mean_probs.dtype) * cdf[:, -1:]
    cls = (cdf[:, None, :] <= t[:, :, None]).sum(axis=2)
    cls = np.minimum(cls, c - 1)
    counts = np.zeros((a, c), dtype=mean_probs.dtype)
    for j in range(c):
        counts[:, j] = (cls == j).sum(axis=1)
    return counts


def mean_covariance(boxes):
    """compute_mean_covariance_tf (:220-244): two-pass, unbiased (N-1)."""
    n = boxes.shape[0]
    mean = boxes.mean(axis=0, keepdims=True)
    d = boxes - mean
    cov = np.einsum("nmi,nmj->mij", d, d) / boxes.dtype.type(n - 1.0)
    return mean[0], cov


def unit_lower_inverse(l_raw):
    """inv(set_diag(L, 1)) for unit-lower 4x4 (:74-78) by forward substitution."""
    m = l_raw.shape[0]
    inv = np.zeros_like(l_raw)
    for i in range(4):
        inv[:, i, i] = 1
        for j in range(i):
            s = np.zeros(m, dtype=l_raw.dtype)
            for k in range(j, i):
                s = s + l_raw[:, i, k] * inv[:, k, j]
            inv[:, i, j] = -s
    return inv


def aleatoric_covariance(covar_raw_mean, use_full_covar):
    """:62-84.  covar_raw_mean [M,4,4] = mean over MC samples of fill_triangular output."""
    diag = np.exp(np.diagonal(covar_raw_mean, axis1=1, axis2=2))     # [M,4]
    var = np.zeros_like(covar_raw_mean)
    for i in range(4):
        var[:, i, i] = diag[:, i]
    if use_full_covar and covar_raw_mean.size != 0:
        l = unit_lower_inverse(covar_raw_mean)
        return l @ var @ np.transpose(l, (0, 2, 1))
    return var


def gaussian_entropy(cov):
    """compute_gaussian_entropy_tf (:247-263)."""
    d = cov.shape[2] / 2.0
    return d + d * np.log(2.0 * np.pi) + 0.5 * np.log(np.linalg.det(cov))


def categorical_entropy(p):
    """compute_categorical_entropy_tf (:266-277)."""
    return -np.sum(p * np.log(p), axis=1)


def bayes_od_posterior(prediction, anchors, uniforms, bayes_od_config, use_full_covar=False,
                       dataset_name="bdd", orig_size=None, net_size=None, dtype=np.float64,
                       return_debug=False):
    """Everything in bayes_od_inference between the model call and the NMS (:25-202).

    prediction: dict with 'anchors_class_predictions' [N,A,C], 'anchors_box_predictions' [N,A,4],
                optional 'anchors_box_covar_predictions' [N,A,4,4]
    anchors   : [A,4] (v,u,h,w);   uniforms: [A,30]
    returns dict: counts (Dirichlet posterior) [M,C], score [M,C], means [M,4,1], covs [M,4,4],
                  ranking [M], keep [A] bool, corners [M,4]
    """
    t = dtype
    cls = prediction["anchors_class_predictions"].astype(t)
    box_t = prediction["anchors_box_predictions"].astype(t)
    anchors = anchors.astype(t)
    n, a, c = cls.shape

    boxes = geometry.box_from_anchor_and_target(anchors[None], box_t)          # :26-29
    probs = softmax(cls)                                                       # :31-32
    mean_probs = probs.mean(axis=0)
    samples = sample_counts(mean_probs, uniforms)                              # :37-46
    keep = np.argmax(samples, axis=1) != (c - 1)                               # :48-51
    like_counts = samples[keep]
    boxes_f = boxes[:, keep]
    mu, cov_epi = mean_covariance(boxes_f)                                     # :59-60

    if "anchors_box_covar_predictions" in prediction:
        raw = prediction["anchors_box_covar_predictions"].astype(t).mean(axis=0)[keep]
        cov_al = aleatoric_covariance(raw, use_full_covar)
    else:
        cov_al = np.zeros_like(cov_epi)
    cov_lik = (t(MIX_ALEATORIC) * cov_al + t(MIX_EPISTEMIC) * cov_epi) / t(11.0)   # :86-87

    if bayes_od_config["dirichlet_prior"]["type"] == "non_informative":        # :90-94
        alpha = t(1.0) / t(c)
        post_counts = like_counts + alpha
    else:
        alpha = None
        post_counts = like_counts
    score = post_counts / post_counts.sum(axis=1, keepdims=True)               # :96-97

    m = int(keep.sum())
    prior_cov = None
    if bayes_od_config["gaussian_prior"]["type"] == "isotropic":               # :100-142
        prec = np.linalg.inv(cov_lik) if m else cov_lik
        pv = t(bayes_od_config["gaussian_prior"]["isotropic_variance"])
        prior_cov = np.tile((np.eye(4, dtype=t) * pv)[None], (m, 1, 1))
        prior_prec = np.tile((np.eye(4, dtype=t) / pv)[None], (m, 1, 1))
        prior_mean = anchors[keep][:, :, None]                                 # :122-123
        post_prec = prec + prior_prec
        post_cov = np.linalg.inv(post_prec) if m else post_prec
        inter = prior_prec @ prior_mean + prec @ mu[:, :, None]
        post_mean = post_cov @ inter
    else:
        post_cov = cov_lik
        post_mean = mu[:, :, None]

    if dataset_name == "kitti":                                                # :147-167
        s = np.asarray(orig_size[:2], dtype=np.float64) / np.asarray(net_size[:2], dtype=np.float64)
        smat = np.diag(np.tile(s, 2)).astype(np.float32).astype(t)[None]
        post_mean = smat @ post_mean
        post_cov = smat @ post_cov @ np.transpose(smat, (0, 2, 1))

    if (bayes_od_config["ranking_method"] == "joint_entropy"
            and bayes_od_config["gaussian_prior"]["type"] != "None"
            and bayes_od_config["dirichlet_prior"]["type"] != "None"):         # :169-200
        g_gain = gaussian_entropy(prior_cov) - gaussian_entropy(post_cov)
        g_gain = (g_gain - g_gain.min()) / max(1.0, g_gain.max() - g_gain.min())
        prior_score = np.full((1, c), alpha, dtype=t)
        prior_score = prior_score / prior_score.sum()
        c_gain = categorical_entropy(prior_score) - categorical_entropy(score)
        c_gain = (c_gain - c_gain.min()) / max(0.001, c_gain.max() - c_gain.min())
        ranking = (c_gain + g_gain).astype(t)
    else:
        ranking = score.max(axis=1) if m else np.zeros((0,), dtype=t)          # :202

    corners = geometry.vuhw_to_vuvu(post_mean[:, :, 0]) if m else np.zeros((0, 4), dtype=t)
    out = {"counts": post_counts, "score": score, "means": post_mean, "covs": post_cov,
           "ranking": ranking, "keep": keep, "corners": corners}
    if return_debug:
        out.update({"mean_probs": mean_probs, "samples": samples, "boxes": boxes,
                    "cov_epi": cov_epi, "cov_al": cov_al, "cov_lik": cov_lik, "mu": mu})
    return out
