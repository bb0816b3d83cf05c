"""On-disk outputs of the inference loop in the reference's formats
(src/retina_net/experiments/validation_utils.py:96-107,183-272; run_inference.py:90-115,174-251)."""
import json
import os

import numpy as np


def strip_checkpoint_id(checkpoint_dir):
    """'.../name-101' -> 101 (validation_utils.py:96-107)."""
    return int(checkpoint_dir.split('-')[-1])


def predictions_to_bdd_format(output_boxes, output_classes, frame_name, category_list):
    """Boxes (v1,u1,v2,u2) -> BDD records with bbox [x1,y1,x2,y2]; background-argmax rows are
    dropped (validation_utils.py:183-214)."""
    out = []
    for box, cls in zip(output_boxes, output_classes):
        k = int(np.argmax(cls))
        if k < len(category_list):
            out.append({"name": frame_name, "timestep": 1000, "category": category_list[k],
                        "bbox": [float(box[1]), float(box[0]), float(box[3]), float(box[2])],
                        "score": cls[k].tolist()})
    return out


def predictions_to_kitti_format(output_boxes, output_classes):
    """KITTI label rows for Car / Pedestrian only, as in the reference (validation_utils.py:217-272)."""
    names = {0: 'Car', 1: 'Pedestrian'}
    rows = []
    for box, cls in zip(output_boxes, output_classes):
        b = np.copy(box[::-1])
        k = int(np.argmax(cls))
        if k in names:
            row = [names[k], -1, -1, -10]
            row.extend(b[2:4]); row.extend(b[0:2])
            row.extend([-10, -10, -10]); row.extend([-10, -10, -10]); row.extend([-10])
            row.append(cls[k])
            rows.append(row)
    return np.asarray(rows)


class PredictionWriter(object):
    """Directory layout predictions/testing/<dataset>/<ckpt_id>/<method>_<fusion>/{data,mean,cov,
    cat_param,cat_count} and per-frame files (run_inference.py:90-115,174-236)."""

    def __init__(self, predictions_dir, dataset, ckpt_id, uncertainty_method='bayes_od', fusion_method='none'):
        self.dataset = dataset
        root = os.path.join(predictions_dir, 'testing', dataset, str(ckpt_id), uncertainty_method)
        if uncertainty_method == 'bayes_od':
            root += '_' + fusion_method
        self.root = root
        self.dirs = {k: os.path.join(root, k) for k in ('data', 'mean', 'cov', 'cat_param', 'cat_count')}
        for d in self.dirs.values():
            os.makedirs(d, exist_ok=True)
        self.results = []

    def write(self, sample_id, output_boxes_vuvu, output_classes_mapped, output_boxes_vuhw, output_covs,
              output_classes, output_counts, category_list=None):
        if self.dataset == 'kitti':
            rows = predictions_to_kitti_format(output_boxes_vuvu, output_classes_mapped)
            path = os.path.join(self.dirs['data'], sample_id + '.txt')
            if rows.size == 0:
                np.savetxt(path, [])
            else:
                np.savetxt(path, rows, newline='\r\n', fmt='%s')
        else:
            self.results.extend(predictions_to_bdd_format(output_boxes_vuvu, output_classes_mapped, sample_id,
                                                          category_list))
        np.save(os.path.join(self.dirs['mean'], sample_id + '.npy'), output_boxes_vuhw)
        np.save(os.path.join(self.dirs['cov'], sample_id + '.npy'), output_covs)
        np.save(os.path.join(self.dirs['cat_param'], sample_id + '.npy'), output_classes)
        np.save(os.path.join(self.dirs['cat_count'], sample_id + '.npy'), output_counts)

    def close(self):
        if self.dataset != 'kitti':
            with open(os.path.join(self.dirs['data'], 'predictions.json'), 'w') as fp:
                json.dump(self.results, fp, indent=4, separators=(',', ': '))
