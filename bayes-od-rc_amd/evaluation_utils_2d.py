"""Offline 2-D detection metrics with the reference's function names, arguments and return values
(src/core/evaluation_utils_2d.py:12-290): PASCAL-VOC style average precision per category
(``evaluate_detection`` -> ``cat_pc`` -> ``get_ap``), the minimum uncertainty error of the entropy ranking
(``evaluate_u_error`` -> ``compute_mu_error``), IoU and entropy helpers.  Host NumPy, like the reference: these
run once per checkpoint over a few thousand json records, far from the hot path.
Inputs are the BDD-format record lists the writers produce: dicts with 'name', 'category', 'bbox' [x1,y1,x2,y2],
'score' (and 'entropy_score' for the uncertainty error)."""
from collections import defaultdict

import numpy as np


def two_d_iou(box, boxes):
    """:12-47 -- IoU of one box against many, +1 pixel convention, rounded to 3 decimals."""
    boxes = np.asarray(boxes, dtype=np.float64)
    iw = np.maximum(np.minimum(box[2], boxes[:, 2]) - np.maximum(box[0], boxes[:, 0]) + 1.0, 0.0)
    ih = np.maximum(np.minimum(box[3], boxes[:, 3]) - np.maximum(box[1], boxes[:, 1]) + 1.0, 0.0)
    inter = iw * ih
    area = (box[2] - box[0] + 1.0) * (box[3] - box[1] + 1.0)
    areas = (boxes[:, 2] - boxes[:, 0] + 1.0) * (boxes[:, 3] - boxes[:, 1] + 1.0)
    iou = np.zeros(len(boxes), np.float64)
    ok = (iw > 0) & (ih > 0)
    iou[ok] = inter[ok] / (area + areas[ok] - inter[ok])
    return iou.round(3)


def group_by_key(detections, key):
    groups = defaultdict(list)
    for d in detections:
        groups[d[key]].append(d)
    return groups


def get_ap(recalls, precisions):
    """:253-269 -- area under the monotone precision envelope."""
    r = np.concatenate(([0.0], np.asarray(recalls, dtype=np.float64), [1.0]))
    p = np.concatenate(([0.0], np.asarray(precisions, dtype=np.float64), [0.0]))
    p = np.maximum.accumulate(p[::-1])[::-1]
    i = np.where(r[1:] != r[:-1])[0]
    return np.sum((r[i + 1] - r[i]) * p[i + 1])


def _match(gt, predictions, thresholds):
    """Greedy TP/FP marking shared by cat_pc (:57-110) and compute_mu_error (:129-193): every prediction claims the
    ground-truth box of its image it overlaps most, once per threshold."""
    image_gts = group_by_key(gt, 'name')
    gt_boxes = {k: np.array([[float(z) for z in b['bbox']] for b in v]) for k, v in image_gts.items()}
    checked = {k: np.zeros((len(v), len(thresholds))) for k, v in image_gts.items()}
    nd = len(predictions)
    tp = np.zeros((nd, len(thresholds)))
    fp = np.zeros((nd, len(thresholds)))
    ious = np.zeros(nd)
    for i, p in enumerate(predictions):
        box = p['bbox']
        ovmax, jmax = -np.inf, -1
        g = gt_boxes.get(p['name'])
        if g is not None and len(g) > 0:
            iw = np.maximum(np.minimum(g[:, 2], box[2]) - np.maximum(g[:, 0], box[0]) + 1.0, 0.0)
            ih = np.maximum(np.minimum(g[:, 3], box[3]) - np.maximum(g[:, 1], box[1]) + 1.0, 0.0)
            inters = iw * ih
            uni = ((box[2] - box[0] + 1.0) * (box[3] - box[1] + 1.0)
                   + (g[:, 2] - g[:, 0] + 1.0) * (g[:, 3] - g[:, 1] + 1.0) - inters)
            overlaps = inters / uni
            ovmax, jmax = np.max(overlaps), int(np.argmax(overlaps))
            ious[i] = ovmax
        for t, thr in enumerate(thresholds):
            if ovmax > thr and checked[p['name']][jmax, t] == 0:
                tp[i, t] = 1.0
                checked[p['name']][jmax, t] = 1
            else:
                fp[i, t] = 1.0
    return tp, fp, ious


def cat_pc(gt, predictions, thresholds):
    """:52-126 -> (recalls, precisions, ap[len(thresholds)], optimal score threshold, maximum f-score)."""
    num_gts = len(gt)
    predictions = sorted(predictions, key=lambda x: x['score'], reverse=True)
    tp, fp, _ = _match(gt, predictions, thresholds)
    fp = np.cumsum(fp, axis=0)
    tp = np.cumsum(tp, axis=0)
    recalls = tp / float(num_gts)
    precisions = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    ap = np.array([get_ap(recalls[:, t], precisions[:, t]) for t in range(len(thresholds))])
    f_score = 2 * (precisions * recalls) / (precisions + recalls + 1e-6)
    best = int(np.argmax(f_score))                   # flat index, used as a row index exactly like the reference
    return recalls, precisions, ap, predictions[best]['score'], f_score.flat[best]


def compute_mu_error(gt, predictions, thresholds):
    """:129-212 -> (minimum uncertainty error, entropy score at the minimum); ranks by 'entropy_score' ascending."""
    predictions = sorted(predictions, key=lambda x: x['entropy_score'], reverse=False)
    tp, fp, ious = _match(gt, predictions, thresholds)
    for i, p in enumerate(predictions):               # the reference annotates the records in place
        p['iou'] = float(ious[i])
        p['is_tp'] = int(tp[i, -1])
    total_tp, total_fp = np.sum(tp, axis=0), np.sum(fp, axis=0)
    fp = np.cumsum(fp, axis=0)
    tp = np.cumsum(tp, axis=0)
    u_error = 0.5 * (total_tp - tp) / np.maximum(total_tp, 1.0) + 0.5 * fp / np.maximum(total_fp, 1.0)
    scores = np.array([p['entropy_score'] for p in predictions])
    return np.min(u_error), scores[int(np.argmin(u_error))]


def evaluate_detection(gt, pred, iou_thresholds=(0.5,)):
    """:215-233 -> (mAP in percent, per-category APs, category list, optimal score thresholds, maximum f-scores)."""
    cat_gt, cat_pred = group_by_key(gt, 'category'), group_by_key(pred, 'category')
    cat_list = sorted(cat_gt.keys())
    aps = np.zeros((len(iou_thresholds), len(cat_list)))
    thr = np.zeros_like(aps)
    fmax = np.zeros_like(aps)
    for i, cat in enumerate(cat_list):
        if cat in cat_pred:
            _, _, ap, t, f = cat_pc(cat_gt[cat], cat_pred[cat], list(iou_thresholds))
            aps[:, i], thr[:, i], fmax[:, i] = ap, t, f
    aps *= 100
    return np.mean(aps), aps.flatten().tolist(), cat_list, thr.flatten().tolist(), fmax.flatten().tolist()


def evaluate_u_error(gt, pred, iou_thresholds=(0.5,)):
    """:236-250 -> (per-category minimum uncertainty errors, their mean, category list, scores at the minima)."""
    cat_gt, cat_pred = group_by_key(gt, 'category'), group_by_key(pred, 'category')
    cat_list = sorted(cat_gt.keys())
    mue = np.zeros((len(iou_thresholds), len(cat_list)))
    at = np.zeros_like(mue)
    for i, cat in enumerate(cat_list):
        if cat in cat_pred:
            mue[:, i], at[:, i] = compute_mu_error(cat_gt[cat], cat_pred[cat], list(iou_thresholds))
    return mue.flatten().tolist(), np.mean(mue), cat_list, at.flatten().tolist()


def compute_gaussian_entropy_np(cov):
    """:280-285"""
    k = cov.shape[1] / 2.0
    det = np.round(np.linalg.det(cov), 5) + 1e-12
    return k + k * np.log(2 * np.pi) + 0.5 * np.log(det)


def compute_categorical_entropy_np(cat_params):
    """:288-290"""
    return -np.sum(cat_params * np.log(cat_params))
