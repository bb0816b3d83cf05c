"""Oracle restatement of the host-side NumPy half of the reference
(SURVEY.md rows a16, a20): ``bayes_od_clustering`` and ``map_dataset_classes``,
src/retina_net/experiments/inference_utils.py:285-364 and :372-404.
PINNED against tests/golden/clustering.npz / class_map.npz (captured by importing the reference).
"""
import numpy as np

COV_CALIBRATION = 70.0     # inference_utils.py:359-361

CATEGORY_IDX = {            # src/core/constants.py:15-29 (data, copied as constants)
    "kitti": {"car": 0, "pedestrian": 1, "cyclist": 2, "bknd": 3},
    "bdd": {"car": 0, "truck": 1, "bus": 2, "person": 3, "rider": 4, "bike": 5, "motor": 6, "bknd": 7},
}
SET_TO_SET = {              # src/core/constants.py:33-43
    "bdd_kitti": {"car": "car", "truck": "car", "bus": "car", "person": "pedestrian",
                  "rider": "cyclist", "bike": "cyclist", "motor": "cyclist", "bknd": "bknd"},
    "coco_rvc": {}, "coco_pascal": {},
}


def kl_rows(p, q):
    """scipy.stats.entropy(p.T, q.T) column-wise: sum p*log(p/q) with both normalised (:344)."""
    p = p / p.sum(axis=1, keepdims=True)
    q = q / q.sum(axis=1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        term = np.where(p > 0, p * np.log(p / q), 0.0)
    return term.sum(axis=1)


def bayes_od_clustering(counts, means, covs, centres, affinity, affinity_threshold=0.7,
                        return_margins=False):
    """:285-364.  counts [M,C], means [M,4,1], covs [M,4,4], centres [K], affinity [M,M]."""
    f_means, f_covs, f_scores, f_counts, margins = [], [], [], [], []
    for centre in centres:
        members = affinity[:, centre] > affinity_threshold                 # :316
        c_means, c_covs = means[members], covs[members]
        precs = np.array([np.linalg.inv(c) for c in c_covs])               # :321-322
        final_cov = np.linalg.inv(np.sum(precs, axis=0))                   # :324
        tmp = np.sum(np.array([p @ m for p, m in zip(precs, c_means)]), axis=0)
        f_covs.append(final_cov)
        f_means.append(final_cov @ tmp)                                    # :327-331
        cnt = counts[members]
        score = cnt / np.expand_dims(np.sum(cnt, axis=1), axis=1)          # :335-336
        margin = np.inf
        if score.shape[0] > 3:                                             # :338-349
            centre_score = counts[centre] / np.sum(counts[centre])
            centre_score = np.repeat(centre_score[None], score.shape[0], axis=0)
            kl = kl_rows(centre_score, score)
            order = np.argsort(kl, kind="stable")
            margin = kl[order[3]] - kl[order[2]]          # 0 => argpartition tie (App. A.11)
            inds = order[:3]
            score, cnt = score[inds], cnt[inds]
        f_scores.append(np.mean(score, axis=0))                            # :351-352
        f_counts.append(np.sum(cnt, axis=0))
        margins.append(margin)
    out = (np.array(f_scores), np.array(f_means),
           np.array(f_covs) * COV_CALIBRATION, np.array(f_counts))
    return out + (np.array(margins),) if return_margins else out


def map_dataset_classes(input_dataset, target_dataset, output_classes):
    """:372-404.  Width of the result is len(target_dict)+1 (= 5 for kitti: the dict already
    contains 'bknd' -- preserved quirk)."""
    mapping = SET_TO_SET[input_dataset + "_" + target_dataset]
    if not mapping:
        return output_classes
    in_d, tg_d = CATEGORY_IDX[input_dataset], CATEGORY_IDX[target_dataset]
    out = np.zeros([output_classes.shape[0], len(tg_d) + 1])
    if output_classes.ndim == 1:
        output_classes = output_classes[:, None]
    max_idx = np.argmax(output_classes, axis=1)
    names = list(in_d.keys())
    idxs = np.take(list(in_d.values()), max_idx)
    mapped = [tg_d[mapping[names[i]]] for i in idxs]
    best = np.amax(output_classes, axis=1)
    for s, mi, row in zip(best, mapped, out):
        row[mi] = s
    return out
