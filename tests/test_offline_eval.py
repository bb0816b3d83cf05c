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


def test_kitti_tree(tmp_path):
    """KITTI twins: label_2 text files + the per-frame text predictions the KITTI writer emits."""
    rng = np.random.default_rng(9)
    label_dir = tmp_path / 'label_2'
    label_dir.mkdir()
    w = PredictionWriter(str(tmp_path), 'kitti', 3)
    names = {0: 'Car', 1: 'Pedestrian'}
    for f in range(3):
        rows, vuvu, cls5, cls8 = [], [], [], []
        for _ in range(2):
            x1, y1 = int(rng.integers(10, 600)), int(rng.integers(10, 200))
            bw, bh = int(rng.integers(40, 120)), int(rng.integers(40, 100))
            c = int(rng.integers(0, 2))
            rows.append('%s 0.00 0 -10 %d.00 %d.00 %d.00 %d.00 1 1 1 0 0 0 0' % (names[c], x1, y1, x1 + bw, y1 + bh))
            vuvu.append([y1, x1, y1 + bh, x1 + bw])
            p5 = np.full(5, 0.02, np.float32); p5[c] = 0.92
            p8 = np.full(8, 0.01, np.float32); p8[0 if c == 0 else 3] = 0.93        # BDD-trained vector: car = 0, person = 3
            cls5.append(p5); cls8.append(p8)
        rows.append('DontCare -1 -1 -10 1.00 1.00 5.00 5.00 -1 -1 -1 -1000 -1000 -1000 -10')
        (label_dir / ('%06d.txt' % f)).write_text('\n'.join(rows) + '\n')
        vuvu = np.array(vuvu, np.float32)
        w.write('%06d' % f, vuvu, np.array(cls5), box_utils.vuvu_to_vuhw_np(vuvu), np.tile(np.eye(4, dtype=np.float32)[None] * 0.004, (2, 1, 1)),
                np.array(cls8), np.array(cls8) * 30)
    w.close()
    g_cls, g_box = offline_eval.read_kitti_labels(str(label_dir / '000000.txt'), 'all', ('car', 'pedestrian'))
    assert g_cls.shape == (2, 4) and g_box.shape == (2, 4)
    gt, pred = offline_eval.kitti_records(str(label_dir), w.root)
    assert len(gt) == 6 and len(pred) == 6
    ap = offline_eval.main(['ap', '--dataset', 'kitti', '--labels', str(label_dir), '--predictions', w.root])
    assert abs(ap['mean_ap'] - 100.0) < 1e-9
    res = offline_eval.main(['pdq', '--dataset', 'kitti', '--labels', str(label_dir), '--predictions', w.root])
    assert res['TP'] == 6 and res['FP'] == 0 and res['FN'] == 0 and res['score'] > 50.0
