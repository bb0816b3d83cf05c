"""CPU: the offline-evaluation drivers (bayes_od_rc_amd/offline_eval.py; reference offline_eval/bdd/compute_ap.py,
compute_uncertainty_error.py, compute_pdq.py) on a small prediction tree written by the package's own
PredictionWriter: detections that reproduce the labels score (near) perfectly, clutter lowers the scores, and the
three reports agree with direct calls of the pinned metric functions."""
import json
import os

import numpy as np

from bayes_od_rc_amd import box_utils, evaluation_utils_2d as ev, offline_eval, prob_detection_quality as pdq
from bayes_od_rc_amd.writers import PredictionWriter

CATS = list(offline_eval.BDD_CATEGORIES)
SHAPE = (96, 128)


def _tree(tmp_path, clutter):
    rng = np.random.default_rng(5)
    gt = []
    w = PredictionWriter(str(tmp_path), 'bdd', 7)
    for f in range(4):
        name = 'frame%02d.jpg' % f
        boxes, onehot = [], []
        for _ in range(0 if f == 3 else 2):                  # the last frame has no labelled object
            x1, y1 = rng.integers(5, 60), rng.integers(5, 40)
            bw, bh = rng.integers(20, 50), rng.integers(20, 40)
            c = int(rng.integers(0, 7))
            gt.append({'name': name, 'category': CATS[c], 'bbox': [float(x1), float(y1), float(x1 + bw), float(y1 + bh)]})
            boxes.append([y1, x1, y1 + bh, x1 + bw])
            p = np.full(8, 0.01, np.float32); p[c] = 0.93
            onehot.append(p)
        for _ in range(clutter):
            x1, y1 = rng.integers(0, 90), rng.integers(0, 60)
            boxes.append([y1, x1, y1 + 25, x1 + 30])
            p = np.full(8, 0.05, np.float32); p[int(rng.integers(0, 7))] = 0.65
            onehot.append(p)
        vuvu = np.array(boxes, np.float32).reshape(-1, 4)
        cls = np.array(onehot, np.float32).reshape(-1, 8)
        vuhw = box_utils.vuvu_to_vuhw_np(vuvu) if len(vuvu) else np.zeros((0, 4), np.float32)
        covs = np.tile(np.eye(4, dtype=np.float32)[None] * 0.004, (len(vuvu), 1, 1))
        w.write(name, vuvu, cls, vuhw, covs, cls, cls * 30, category_list=CATS)
    w.close()
    labels = os.path.join(str(tmp_path), 'labels.json')
    with open(labels, 'w') as fp:
        json.dump(gt, fp)
    return w.root, labels, gt


def test_reports_on_exact_detections(tmp_path):
    root, labels, gt = _tree(tmp_path, clutter=0)
    ap = offline_eval.main(['ap', '--labels', labels, '--predictions', root])
    assert abs(ap['mean_ap'] - 100.0) < 1e-9 and ap['out_of_distribution_ratio'] == 0.0
    mue = offline_eval.main(['mue', '--labels', labels, '--predictions', root, '--entropy', 'categorical'])
    assert mue['mean_mue'] <= 0.5 and len(mue['mue']) == len(mue['categories'])
    res = offline_eval.main(['pdq', '--labels', labels, '--predictions', root, '--image-size', str(SHAPE[0]), str(SHAPE[1])])
    assert res['TP'] == 6 and res['FP'] == 0 and res['FN'] == 0
    assert res['score'] > 50.0 and 0.9 < res['avg_label_quality'] < 0.94


def test_clutter_lowers_scores_and_matches_direct_calls(tmp_path):
    root, labels, gt = _tree(tmp_path, clutter=3)
    with open(os.path.join(root, 'data', 'predictions.json')) as fp:
        pred = json.load(fp)
    ap = offline_eval.ap_report(gt, pred)
    direct = ev.evaluate_detection(gt, pred, iou_thresholds=[0.5])
    assert ap['ap'] == [float(a) for a in direct[1]] and ap['categories'] == list(direct[2])
    frames = sorted(f[:-4] for f in os.listdir(os.path.join(root, 'mean')))
    res = offline_eval.pdq_report(gt, root, frames, SHAPE)
    clean_root, _, gt2 = _tree(tmp_path / 'clean', clutter=0)
    clean = offline_eval.pdq_report(gt2, clean_root, frames, SHAPE)
    assert res['FP'] > 0 and res['score'] < clean['score']
    # the frame without labels enters as one background-class unit box that is never counted
    onehot, boxes = offline_eval.read_bdd_frame('frame03.jpg', gt)
    assert onehot.shape == (1, 8) and onehot[0, 7] == 1 and boxes.tolist() == [[0.0, 0.0, 1.0, 1.0]]
    gts, dets = pdq.frame_instances(onehot, boxes, np.load(os.path.join(root, 'mean', 'frame03.jpg.npy')),
                                    np.load(os.path.join(root, 'cov', 'frame03.jpg.npy')),
                                    np.load(os.path.join(root, 'cat_param', 'frame03.jpg.npy')), SHAPE)
    assert len(gts) == 1 and not pdq.gt_counts_for_pdq(gts[0]) and len(dets) == 3
