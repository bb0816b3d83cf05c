"""Offline evaluation drivers over the prediction tree ``run_inference`` writes (SURVEY.md section 8 row f4;
reference: src/retina_net/offline_eval/bdd/compute_ap.py:55-78, compute_uncertainty_error.py:60-137,
compute_pdq.py:64-160 -- the KITTI twins differ only in how the label files are read).

The reference scripts hard-code their paths and print a table; here the same computations are functions of
(ground-truth records, prediction tree) plus a small CLI:

    python -m bayes_od_rc_amd.offline_eval ap  --labels val.json --predictions <..>/bayes_od_none
    python -m bayes_od_rc_amd.offline_eval mue --labels val.json --predictions <..>/bayes_od_none --entropy gaussian
    python -m bayes_od_rc_amd.offline_eval pdq --labels val.json --predictions <..>/bayes_od_none --image-size 720 1280

    python -m bayes_od_rc_amd.offline_eval ap|pdq --dataset kitti --labels <label_2 dir> --predictions <..>/bayes_od_none

``--predictions`` is the directory holding ``data/ mean/ cov/ cat_param/`` (run_inference.py:90-115); labels are
BDD-format records ``{name, category, bbox: [x1, y1, x2, y2]}`` or, for KITTI, the label_2 text files (converted to the
same records by ``kitti_records``).  Host NumPy like the reference.
"""
import argparse
import json
import os

import numpy as np

from . import evaluation_utils_2d as ev
from . import prob_detection_quality as pdq
from .box_utils import vuhw_to_vuvu_np

BDD_CATEGORIES = ('car', 'truck', 'bus', 'person', 'rider', 'bike', 'motor')


def ap_report(gt_records, prediction_records, iou_threshold=0.5):
    """compute_ap.py:55-77: per-category AP at one IoU threshold, means over the categories that have ground truth
    (AP > 0), and the share of predictions whose category the label set does not contain."""
    mean_ap, aps, cat_list, opt_thr, max_f = ev.evaluate_detection(gt_records, prediction_records, iou_thresholds=[iou_threshold])
    aps_a = np.array(aps)
    seen = aps_a > 0.0
    ood = np.array([0 if p['category'] in cat_list else 1 for p in prediction_records])
    return {'mean_ap': float(np.mean(aps_a[seen])) if seen.any() else 0.0,
            'mean_max_f_score': float(np.mean(np.array(max_f)[seen])) if seen.any() else 0.0,
            'mean_optimal_score_threshold': float(np.mean(np.array(opt_thr)[seen])) if seen.any() else 0.0,
            'out_of_distribution_ratio': float(np.sum(ood) / max(len(prediction_records), 1)),
            'categories': list(cat_list), 'ap': [float(a) for a in aps]}


def _load(tree, sub, frame):
    return np.load(os.path.join(tree, sub, frame) + '.npy')


def entropy_ranked_predictions(tree, frames, category_names=BDD_CATEGORIES + ('bkgrnd',), entropy_method='gaussian',
                               compute_method='Categorical'):
    """compute_uncertainty_error.py:84-128: one record per detection with its class (arg-max of the categorical
    parameters) and the entropy it is ranked by."""
    records = []
    for frame in frames:
        means = _load(tree, 'mean', frame)
        if not means.size:
            continue
        boxes = vuhw_to_vuvu_np(means)
        cats = _load(tree, 'cat_param', frame)
        covs = _load(tree, 'cov', frame)
        names = [category_names[i] for i in np.argmax(cats, axis=1)]
        if entropy_method == 'gaussian':
            ent = [ev.compute_gaussian_entropy_np(c) for c in covs]
        elif entropy_method == 'categorical':
            ent = [ev.compute_categorical_entropy_np(c) for c in cats]
        else:
            raise ValueError('Invalid entropy method: %s' % entropy_method)
        for b, name, e in zip(boxes, names, ent):
            b = b.tolist()
            records.append({'name': frame, 'category': 'All' if compute_method == 'All' else name,
                            'bbox': [b[1], b[0], b[3], b[2]], 'entropy_score': e})
    return records


def uncertainty_error_report(gt_records, tree, frames, category_names=BDD_CATEGORIES + ('bkgrnd',), entropy_method='gaussian',
                             compute_method='Categorical', iou_threshold=0.5):
    """compute_uncertainty_error.py:66-137: minimum uncertainty error per category and its mean."""
    gt = [dict(g) for g in gt_records if g['category'] in category_names]
    if compute_method == 'All':
        for g in gt:
            g['category'] = 'All'
    pred = entropy_ranked_predictions(tree, frames, category_names, entropy_method, compute_method)
    mues, mean_mue, cat_list, _ = ev.evaluate_u_error(gt, pred, iou_thresholds=[iou_threshold])
    return {'mean_mue': float(mean_mue), 'categories': list(cat_list), 'mue': [float(m) for m in mues]}


def read_bdd_frame(frame, gt_records, categories=BDD_CATEGORIES):
    """One frame's (one-hot classes [G, C+1], boxes [G, 4] as x1 y1 x2 y2); a frame without labelled objects yields one
    background-class unit box, as the reference's reader does (demos/demo_utils/bdd_demo_utils.py:4-61, pdq_eval=True):
    it is too small to count but takes part in the assignment."""
    rows = [g for g in gt_records if g['name'] == frame and g['category'] in categories]
    if not rows:
        onehot = np.zeros((1, len(categories) + 1), np.float32)
        onehot[0, len(categories)] = 1
        return onehot, np.array([[0.0, 0.0, 1.0, 1.0]], np.float32)
    onehot = np.zeros((len(rows), len(categories) + 1), np.float32)
    for k, g in enumerate(rows):
        onehot[k, categories.index(g['category'].lower())] = 1
    return onehot, np.array([g['bbox'] for g in rows], np.float32)


# ---------------------------------------------------------------------------------------------------------------
# KITTI twins (offline_eval/kitti/compute_ap.py:40-80, compute_pdq.py:59-125): labels are label_2 text files, the
# box predictions are the per-frame text files run_inference writes into data/ (validation_utils.py:217-272).
# ---------------------------------------------------------------------------------------------------------------
_KITTI_ONEHOT = {'car': [1, 0, 0, 0], 'pedestrian': [0, 1, 0, 0], 'person_sitting': [0, 1, 0, 0], 'cyclist': [0, 0, 1, 0]}


def _kitti_rows(path, ncols):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                           # (an empty prediction file)
        rows = np.loadtxt(path, delimiter=' ', dtype=str, usecols=np.arange(ncols), ndmin=2)
    return rows


def read_kitti_labels(label_path, difficulty='hard', categories=('car', 'pedestrian', 'cyclist')):
    """(one-hot [G, 4], boxes [G, 4] as v1 u1 v2 u2) of the objects passing the difficulty filter
    (demos/demo_utils/kitti_demo_utils.py:8-68); empty arrays when nothing passes."""
    from .datasets import KITTI_DIFF_DICTS, kitti_labels_to_boxes_2d
    rows = _kitti_rows(label_path, 15)
    d = KITTI_DIFF_DICTS[difficulty.lower()]
    if rows.size:
        heights = rows[:, 7].astype(np.float32) - rows[:, 5].astype(np.float32)
        keep = (np.asarray([c.lower() in categories for c in rows[:, 0]], dtype=bool) & (heights >= d['min_height'])
                & (rows[:, 1].astype(np.float64) <= d['max_truncation']) & (rows[:, 2].astype(np.float64) <= d['max_occlusion']))
        rows = rows[keep]
    if not rows.size:
        return np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32)
    onehot = [_KITTI_ONEHOT[c.lower()] for c in rows[:, 0] if c.lower() in _KITTI_ONEHOT]
    return np.array(onehot, np.float32), kitti_labels_to_boxes_2d(rows).astype(np.float32)


def read_kitti_predictions(prediction_path, categories=('car', 'pedestrian', 'cyclist', 'dontcare')):
    """(one-hot [D, 4], boxes [D, 4] as v1 u1 v2 u2, scores [D]) of one frame's prediction file (:71-127)."""
    from .datasets import kitti_labels_to_boxes_2d
    rows = _kitti_rows(prediction_path, 16)
    if rows.size:
        rows = rows[np.asarray([c.lower() in categories for c in rows[:, 0]], dtype=bool)]
    if not rows.size:
        return np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32), np.zeros((0,), np.float32)
    onehot = [_KITTI_ONEHOT[c.lower()] for c in rows[:, 0] if c.lower() in _KITTI_ONEHOT]
    return np.array(onehot, np.float32), kitti_labels_to_boxes_2d(rows).astype(np.float32), rows[:, 15].astype(np.float32)


def kitti_records(label_dir, tree, difficulty='all', categories=('car', 'pedestrian')):
    """(ground-truth records, prediction records) in the BDD record form, one 'name' per frame index, frames where
    either side is empty skipped -- exactly what kitti/compute_ap.py:40-72 hands to evaluate_detection."""
    gt, pred = [], []
    frames = sorted(os.listdir(os.path.join(tree, 'data')))
    for idx, frame in enumerate(frames):
        fid = int(frame[0:6])
        g_cls, g_box = read_kitti_labels(os.path.join(label_dir, '%06d.txt' % fid), difficulty, categories)
        p_cls, p_box, p_score = read_kitti_predictions(os.path.join(tree, 'data', '%06d.txt' % fid), categories)
        if not (g_box.size and p_box.size):
            continue
        for c, b in zip(g_cls, g_box):
            gt.append({'name': str(idx), 'category': categories[int(np.argmax(c))], 'bbox': [float(b[1]), float(b[0]), float(b[3]), float(b[2])], 'score': 1})
        for c, b, sc in zip(p_cls, p_box, p_score):
            pred.append({'name': str(idx), 'category': categories[int(np.argmax(c))], 'bbox': [float(b[1]), float(b[0]), float(b[3]), float(b[2])],
                         'score': float(sc)})
    return gt, pred


def kitti_pdq_report(label_dir, tree, difficulty='all', categories=('car', 'pedestrian'), img_shape=(375, 1300)):
    """kitti/compute_pdq.py:59-135: every frame of the tree in ONE evaluation."""
    matches = []
    for frame in sorted(os.listdir(os.path.join(tree, 'mean'))):
        fid = int(frame[0:6])
        g_cls, g_box = read_kitti_labels(os.path.join(label_dir, '%06d.txt' % fid), difficulty, categories)
        name = '%06d' % fid
        matches.append(pdq.frame_instances(g_cls, g_box, _load(tree, 'mean', name), _load(tree, 'cov', name), _load(tree, 'cat_param', name),
                                           tuple(img_shape), score_threshold=0.5, class_columns=(0, 3), gt_boxes_vuvu=True, clip_max=1300))
    return pdq.evaluate(matches)


def pdq_report(gt_records, tree, frames, img_shape, categories=BDD_CATEGORIES, chunk=1000, score_threshold=0.5445):
    """compute_pdq.py:64-160: PDQ per chunk of 1000 frames, chunk scores averaged, counts summed.  Frames whose
    prediction files are empty are skipped (as the reference does)."""
    rows = []
    for lo in range(0, len(frames), chunk):
        matches = []
        for frame in frames[lo:lo + chunk]:
            covs = _load(tree, 'cov', frame)
            if not covs.size:
                continue
            onehot, boxes = read_bdd_frame(frame, gt_records, categories)
            matches.append(pdq.frame_instances(onehot, boxes, _load(tree, 'mean', frame), covs, _load(tree, 'cat_param', frame),
                                               tuple(img_shape), score_threshold=score_threshold))
        if matches:
            rows.append(pdq.evaluate(matches))
    if not rows:
        return {'score': 0.0, 'TP': 0, 'FP': 0, 'FN': 0, 'avg_spatial_quality': 0.0, 'avg_label_quality': 0.0, 'avg_overall_quality': 0.0}
    out = {k: float(np.mean([r[k] for r in rows])) for k in ('score', 'avg_spatial_quality', 'avg_label_quality', 'avg_overall_quality')}
    out.update({k: int(sum(r[k] for r in rows)) for k in ('TP', 'FP', 'FN')})
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('metric', choices=('ap', 'mue', 'pdq'))
    ap.add_argument('--labels', required=True, help='BDD-format ground-truth json, or (with --dataset kitti) the label_2 directory')
    ap.add_argument('--dataset', default='bdd', choices=('bdd', 'kitti'))
    ap.add_argument('--difficulty', default='all', choices=('easy', 'moderate', 'hard', 'all'))
    ap.add_argument('--predictions', required=True, help='directory with data/ mean/ cov/ cat_param/')
    ap.add_argument('--entropy', default='gaussian', choices=('gaussian', 'categorical'))
    ap.add_argument('--compute-method', default='Categorical', choices=('Categorical', 'All'))
    ap.add_argument('--image-size', type=int, nargs=2, default=(720, 1280), metavar=('H', 'W'))
    args = ap.parse_args(argv)
    if args.dataset == 'kitti':
        if args.metric == 'ap':
            out = ap_report(*kitti_records(args.labels, args.predictions, args.difficulty))
        elif args.metric == 'pdq':
            out = kitti_pdq_report(args.labels, args.predictions, args.difficulty)
        else:
            raise SystemExit('mue --dataset kitti: convert the labels to records with kitti_records() and call uncertainty_error_report()')
        print(json.dumps(out, indent=1))
        return out
    with open(args.labels) as fp:
        gt = json.load(fp)
    if args.metric == 'ap':
        with open(os.path.join(args.predictions, 'data', 'predictions.json')) as fp:
            out = ap_report(gt, json.load(fp))
    else:
        frames = sorted(f[:-4] for f in os.listdir(os.path.join(args.predictions, 'mean')) if f.endswith('.npy'))
        if args.metric == 'mue':
            out = uncertainty_error_report(gt, args.predictions, frames, entropy_method=args.entropy, compute_method=args.compute_method)
        else:
            out = pdq_report(gt, args.predictions, frames, args.image_size)
    print(json.dumps(out, indent=1))
    return out


if __name__ == '__main__':
    main()
