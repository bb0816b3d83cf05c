"""Probability-based Detection Quality (PDQ) of probabilistic box detections -- the offline metric of SURVEY.md
section 8 row f4 (reference: src/retina_net/offline_eval/pdq.py:11-471 and pdq_data_holders.py:13-268, the
variant of the Robotic Vision Challenge metric this code base carries: false positives contribute a quality of
their own, see ``image_quality``).  CPU NumPy / SciPy like the reference; it consumes the per-frame ``mean`` /
``cov`` / ``cat_param`` arrays ``run_inference`` writes (offline_eval/bdd/compute_pdq.py:73-122).

Everything is restated from the definitions:

* a detection is a box whose two corners are bivariate Gaussians; the probability that a pixel belongs to it is
  P(top-left corner is above-left of the pixel) * P(bottom-right corner is below-right of it)  (``PBoxDetInst``);
* spatial quality of a (ground truth, detection) pair = exp of the mean log-loss over the object's pixels, where
  foreground pixels are charged log p and pixels outside the ground-truth box log(1 - p)  (``pair_qualities``);
* label quality = the probability the detection gives to the true class; pair quality = their geometric mean;
* pairs are assigned one-to-one by the Hungarian method on 1 - quality; PDQ = total quality / (TP + FP + FN).

Pinned by golden vectors captured from the reference's own functions (tests/golden/pdq.npz, generator
tests/golden/make_golden.py; the generator aliases the ``np.int`` / ``np.bool`` names NumPy 2 removed).
"""
import numpy as np
from scipy.optimize import linear_sum_assignment
from scipy.spatial.distance import cdist
from scipy.stats import multivariate_normal

HEATMAP_FLOOR = 0.0027          # probabilities below this are treated as 0 (pdq_data_holders.py:8)
ROI_MAHALANOBIS = 3.439         # radius, in standard deviations, of the region the corner CDF is evaluated on (:9)
TINY = 1e-14                    # (:10, pdq.py:8)


def mask_bounding_box(mask):
    """[xmin, ymin, xmax, ymax] (inclusive) of the True pixels (pdq_data_holders.py:270-283)."""
    cols, rows = np.any(mask, axis=0), np.any(mask, axis=1)
    if not cols.any() and not rows.any():
        raise ValueError("No positive pixels found, cannot compute bounding box")
    return [int(np.argmax(cols)), int(np.argmax(rows)),
            int(len(cols) - 1 - np.argmax(cols[::-1])), int(len(rows) - 1 - np.argmax(rows[::-1]))]


class GroundTruthInstance(object):
    """pdq_data_holders.py:13-48 (same constructor and attribute names)."""

    def __init__(self, segmentation_mask, true_class_label, image_id, instance_id, bounding_box=None, num_pixels=None):
        self.segmentation_mask = segmentation_mask
        self.class_label = true_class_label
        self.image_id = image_id
        self.instance_id = instance_id
        has_box = bounding_box is not None and len(bounding_box) > 0
        self.bounding_box = bounding_box if has_box else mask_bounding_box(segmentation_mask)
        self.num_pixels = num_pixels if (num_pixels is not None and num_pixels > 0) else int(np.count_nonzero(segmentation_mask))
        b = self.bounding_box
        self.num_bbox_pixels = (b[2] + 1 - b[0]) * (b[3] + 1 - b[1])


class DetectionInstance(object):
    """pdq_data_holders.py:51-78."""

    def __init__(self, class_list, heatmap=None):
        self._heatmap = heatmap
        self.class_list = class_list

    def calc_heatmap(self, img_size):
        return self._heatmap

    def get_max_class(self):
        return np.argmax(self.class_list)

    def get_max_score(self):
        return np.amax(self.class_list)


class BBoxDetInst(DetectionInstance):
    """Plain box [x1, y1, x2, y2] with one spatial probability; fractional borders are weighted by the covered
    fraction of the border pixel (pdq_data_holders.py:81-117)."""

    def __init__(self, class_list, box, pos_prob=1.0):
        super(BBoxDetInst, self).__init__(class_list)
        self.box = box
        self.pos_prob = pos_prob

    def calc_heatmap(self, img_size):
        h, w = img_size
        out = np.zeros(img_size, dtype=np.float32)
        x1, y1, x2, y2 = self.box
        x1c, y1c = (int(v) for v in np.ceil(self.box[0:2]))
        x2f, y2f = (int(v) for v in np.floor(self.box[2:]))
        left, top, right, bottom = x1c - 1, y1c - 1, x2f + 1, y2f + 1       # border pixels, always present
        rows = slice(max(top, 0), min(bottom + 1, h))
        cols = slice(max(left, 0), min(right + 1, w))
        out[rows, cols] = self.pos_prob
        if top >= 0:
            out[top, cols] *= y1c - y1
        if bottom < h:
            out[bottom, cols] *= y2 - y2f
        if left >= 0:
            out[rows, left] *= x1c - x1
        if right < w:
            out[rows, right] *= x2 - x2f
        return out


class PBoxDetInst(DetectionInstance):
    """Probabilistic box: corners [x1, y1, x2, y2] and their 2x2 covariances [[var_x, c], [c, var_y]]
    (pdq_data_holders.py:120-161)."""

    def __init__(self, class_list, box, covs):
        super(PBoxDetInst, self).__init__(class_list)
        self.box = box
        self.covs = covs

    def calc_heatmap(self, img_size):
        h, w = img_size
        yx = [np.flipud(np.fliplr(c)) for c in self.covs]               # (y, x) order, like the pixel grid
        p_tl = corner_heatmap(img_size, [self.box[1], self.box[0]], yx[0])
        # the bottom-right corner is the top-left corner of the image turned by 180 degrees
        p_br = corner_heatmap(img_size, [h - (self.box[3] + 1), w - (self.box[2] + 1)], np.array(yx[1]).T)
        heat = p_tl * np.fliplr(np.flipud(p_br))
        heat[heat > 1] = 1
        heat[heat < HEATMAP_FLOOR] = 0
        return heat


def corner_roi(img_size, mean, cov):
    """Pixels [xmin, ymin, xmax, ymax] around a Gaussian corner (mean and cov in (y, x) order) on which its CDF moves
    between ~0 and ~1: inside five standard deviations AND within ROI_MAHALANOBIS of the mean, the distance being taken
    at the pixel corner facing the mean (pdq_data_holders.py:164-220)."""
    h, w = img_size
    sy, sx = cov[0, 0] ** 0.5, cov[1, 1] ** 0.5
    x0, y0 = int(max(mean[1] - 5 * sx, 0)), int(max(mean[0] - 5 * sy, 0))
    x1, y1 = int(min(mean[1] + 5 * sx, w - 1)), int(min(mean[0] + 5 * sy, h - 1))
    if abs(np.linalg.det(cov)) < 1e-8:                                   # singular: the coarse box is all there is
        return x0, y0, max(0, x1), max(0, y1)
    nh, nw = max(y1 + 1 - y0, 1), max(x1 + 1 - x0, 1)
    yy, xx = np.mgrid[y0:y0 + nh, x0:x0 + nw]
    pts = np.stack([yy.ravel(), xx.ravel()], axis=1)
    dist = cdist(pts, np.array([mean]), metric='mahalanobis', VI=np.linalg.inv(cov)).reshape(nh, nw)
    my = max(min(int(mean[0] - y0), h - 1), 0)
    mx = max(min(int(mean[1] - x0), w - 1), 0)
    if 0 < my < h - 1:
        dist[:my, :] = dist[1:my + 1, :].copy()
    if 0 < mx < w - 1:
        dist[:, :mx] = dist[:, 1:mx + 1].copy()
    near = dist <= ROI_MAHALANOBIS
    near[my, mx] = True
    bx = mask_bounding_box(near)
    return [max(0, v) for v in (bx[0] + x0, bx[1] + y0, bx[2] + x0, bx[3] + y0)]


def corner_heatmap(img_size, mean, cov):
    """P(the Gaussian corner lies above-left of pixel (y, x)'s lower-right corner, inside the image) for every pixel:
    the CDF on the region of interest, continued constant below / right of it (1 in the far quadrant), minus the mass
    that falls outside the image when the region touches its top or left edge (pdq_data_holders.py:223-268)."""
    h, w = img_size
    heat = np.zeros(img_size, dtype=np.float32)
    g = multivariate_normal(mean=mean, cov=cov, allow_singular=True)
    x0, y0, x1, y1 = corner_roi(img_size, mean, cov)

    def cdf(ys, xs):
        pts = np.dstack(np.mgrid[ys, xs]) - TINY
        return np.asarray(g.cdf(pts)).reshape(pts.shape[0], pts.shape[1])

    heat[y0:y1 + 1, x0:x1 + 1] = cdf(slice(y0 + 1, y1 + 2), slice(x0 + 1, x1 + 2))
    heat[y1:, x0:x1 + 1] = heat[y1, x0:x1 + 1][None, :]
    heat[y0:y1 + 1, x1:] = heat[y0:y1 + 1, x1][:, None]
    heat[y1 + 1:, x1 + 1:] = 1.0
    if x0 == 0:                                                          # mass left of the image
        left = np.zeros((h, 1), dtype=np.float32)
        left[y0:y1 + 1, 0] = cdf(slice(y0 + 1, y1 + 2), slice(0, 1))[:, 0]
        left[y1 + 1:, 0] = left[y1, 0]
        heat -= left
    if y0 == 0:                                                          # mass above the image
        above = np.zeros((1, w), dtype=np.float32)
        above[0, x0:x1 + 1] = cdf(slice(0, 1), slice(x0 + 1, x1 + 2))[0]
        above[0, x1 + 1:] = above[0, x1]
        heat -= above
    if x0 == 0 and y0 == 0:                                              # (counted twice above)
        heat += g.cdf([[[0 - TINY, 0 - TINY]]])
    heat[heat < HEATMAP_FLOOR] = 0
    return heat


def _log(p):
    return np.log(p + TINY)


def gt_counts_for_pdq(gt):
    """Objects of at most 10 px side or 100 px area are ignored (pdq.py:460-471)."""
    b = gt.bounding_box
    return (b[2] - b[0] > 10) and (b[3] - b[1] > 10) and np.count_nonzero(gt.segmentation_mask) > 100


def pair_qualities(gt_instances, det_instances):
    """(overall, spatial, label) quality matrices [G, D] of one image plus the stacked heatmaps [H, W, D] and class
    probabilities [D, C] (pdq.py:157-308)."""
    shape = gt_instances[0].segmentation_mask.shape
    fg = np.stack([g.segmentation_mask for g in gt_instances], axis=2)                   # [H, W, G]
    bg = np.ones(shape + (len(gt_instances),), dtype=bool)
    for k, g in enumerate(gt_instances):
        b = g.bounding_box
        bg[b[1]:b[3] + 1, b[0]:b[2] + 1, k] = False
    n_fg = np.array([[g.num_pixels] for g in gt_instances], dtype=np.int64)              # [G, 1]
    labels = np.array([g.class_label for g in gt_instances], dtype=np.int64)
    probs = np.stack([d.class_list for d in det_instances], axis=0)                      # [D, C]
    heat = np.stack([d.calc_heatmap(shape) for d in det_instances], axis=2)              # [H, W, D]
    fg_loss = np.tensordot(fg, _log(heat), axes=([0, 1], [0, 1]))                        # [G, D]
    bg_loss = np.tensordot(bg, _log(1 - heat) * (heat > 0), axes=([0, 1], [0, 1]))
    spatial = np.exp((fg_loss + bg_loss) / n_fg)
    spatial[np.isclose(spatial, 0)] = 0
    spatial[np.isclose(spatial, 1)] = 1
    label = probs[:, labels].T.astype(np.float32)
    with np.errstate(divide='ignore'):
        overall = np.exp(0.5 * (np.log(label) + np.log(spatial)))                        # geometric mean (0 if either is 0)
    return overall, spatial, label, heat, probs


def image_quality(gt_instances, det_instances):
    """Sums for one image: {'overall', 'spatial', 'label', 'TP', 'FP', 'FN'} (pdq.py:311-452).  Optimal one-to-one
    assignment on 1 - overall quality; assigned pairs of positive quality are true positives (unless the object is
    too small: then the pair is dropped), the rest are false negatives / false positives.  This code base's variant
    also credits every false positive with gmean(exp(mean log(1 - p) over its own box), 1 - max class probability)."""
    n_gt, n_det = len(gt_instances), len(det_instances)
    if n_gt == 0 or n_det == 0:
        return {'overall': 0.0, 'spatial': 0.0, 'label': 0.0, 'TP': 0, 'FP': n_det,
                'FN': int(sum(1 for g in gt_instances if gt_counts_for_pdq(g)))}
    overall, spatial, label, heat, probs = pair_qualities(gt_instances, det_instances)
    n = max(n_gt, n_det)
    q = {k: np.zeros((n, n), dtype=np.float32) for k in ('overall', 'spatial', 'label')}
    q['overall'][:n_gt, :n_det] = overall
    q['spatial'][:n_gt, :n_det] = spatial
    q['label'][:n_gt, :n_det] = label
    cost = np.ones((n, n), dtype=np.float32) - q['overall']
    rows, cols = linear_sum_assignment(cost)
    # (the reference converts quality -> cost -> quality in float32; keep its rounding)
    q = {k: 1 - (np.ones((n, n), dtype=np.float32) - v) for k, v in q.items()}
    tp = fp = fn = 0
    fp_cols = []
    for r, c in zip(rows, cols):
        counted = r < n_gt and gt_counts_for_pdq(gt_instances[r])
        if q['overall'][r, c] > 0:
            if counted:
                tp += 1
            else:
                q['overall'][r, c] = 0.0
        else:
            if counted:
                fn += 1
            if c < n_det:
                fp += 1
                fp_cols.append(c)
    tp_overall = np.sum(q['overall'][rows, cols])
    q['spatial'][q['overall'] == 0] = 0.0
    q['label'][q['overall'] == 0] = 0.0
    tp_spatial = np.sum(q['spatial'][rows, cols])
    tp_label = np.sum(q['label'][rows, cols])
    fp_label = np.array([1.0 - np.max(probs[c]) for c in fp_cols])
    fp_spatial_sum = fp_overall_sum = 0.0
    if fp_label.size:
        maps = np.array([heat[:, :, c] for c in fp_cols])
        area = np.array([(det_instances[c].box[3] - det_instances[c].box[1]) * (det_instances[c].box[2] - det_instances[c].box[0])
                         for c in fp_cols])
        fp_spatial = np.exp(np.sum(_log(1 - maps) * (maps > 0), axis=(1, 2)) / area)
        fp_spatial_sum = np.sum(fp_spatial)
        with np.errstate(divide='ignore'):
            fp_overall_sum = np.sum(np.exp(0.5 * (np.log(fp_spatial) + np.log(fp_label))))
    return {'overall': tp_overall + fp_overall_sum, 'spatial': tp_spatial + fp_spatial_sum,
            'label': tp_label + np.sum(fp_label), 'TP': tp, 'FP': fp, 'FN': fn}


class PDQ(object):
    """Accumulator with the reference's method names (pdq.py:11-142).  ``score`` evaluates the images one after the
    other (the reference fans them out over a multiprocessing pool; the sums are the same)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self._tot_overall_quality = self._tot_spatial_quality = self._tot_label_quality = 0.0
        self._tot_TP = self._tot_FP = self._tot_FN = 0

    def add_img_eval(self, gt_instances, det_instances):
        r = image_quality(gt_instances, det_instances)
        self._tot_overall_quality += r['overall']
        self._tot_spatial_quality += r['spatial']
        self._tot_label_quality += r['label']
        self._tot_TP += r['TP']
        self._tot_FP += r['FP']
        self._tot_FN += r['FN']

    def score(self, matches):
        self.reset()
        for gt_instances, det_instances in matches:
            self.add_img_eval(gt_instances, det_instances)
        return self.get_pdq_score()

    def get_pdq_score(self):
        return self._tot_overall_quality / (self._tot_TP + self._tot_FP + self._tot_FN)

    def _per_detection(self, total):
        n = self._tot_TP + self._tot_FP
        return total / float(n) if n > 0 else 0.0

    def get_avg_spatial_score(self):
        return self._per_detection(self._tot_spatial_quality)

    def get_avg_label_score(self):
        return self._per_detection(self._tot_label_quality)

    def get_avg_overall_quality_score(self):
        return self._per_detection(self._tot_overall_quality)

    def get_assignment_counts(self):
        return self._tot_TP, self._tot_FP, self._tot_FN


# vuhw covariance -> corner covariance: corners (u1, v1, u2, v2) = T (v, u, h, w)  (offline_eval/bdd/compute_pdq.py:91-101)
_VUHW_TO_CORNERS = np.array([[0, 1, 0, -0.5], [1, 0, -0.5, 0], [0, 1, 0, 0.5], [1, 0, 0.5, 0]], dtype=np.float64)


def frame_instances(gt_classes_onehot, gt_boxes_xyxy, pred_means_vuhw, pred_covs, pred_cat_params, img_shape,
                    score_threshold=0.5445, cov_scale=70.0, class_columns=None, gt_boxes_vuvu=False, clip_max=None):
    """One frame of the compute_pdq drivers: ground-truth boxes become box-shaped masks, predictions above
    ``score_threshold`` become PBoxDetInst with the corner covariances T cov T^T * 70 (the x70 of the drivers comes on
    top of the one bayes_od_clustering applied).  Defaults = the BDD driver (offline_eval/bdd/compute_pdq.py:83-140);
    the KITTI driver (kitti/compute_pdq.py:66-118) is ``img_shape=(375, 1300), score_threshold=0.5,
    class_columns=(0, 3)`` (car and person out of the BDD-trained class vector), ``gt_boxes_vuvu=True`` (its label
    reader returns v1 u1 v2 u2) and ``clip_max=1300``.  Returns (gt_instances, det_instances)."""
    from .box_utils import vuhw_to_vuvu_np
    gts = []
    for onehot, box in zip(gt_classes_onehot, gt_boxes_xyxy):
        idx = np.asarray(box).astype(np.int32)
        if gt_boxes_vuvu:
            idx = np.array([idx[1], idx[0], idx[3], idx[2]])
        if clip_max is not None:
            idx = np.clip(idx, 0.0, clip_max).astype(np.int32)
        mask = np.zeros(img_shape, dtype=bool)
        mask[idx[1]:idx[3], idx[0]:idx[2]] = True
        label = int(np.argmax(onehot)) if gt_boxes_vuvu else int(np.where(np.asarray(onehot) == 1)[0].item(0))
        gts.append(GroundTruthInstance(mask, label, 0, 0, bounding_box=idx))
    dets = []
    if np.asarray(pred_covs).size:
        covs = np.matmul(np.matmul(_VUHW_TO_CORNERS, np.asarray(pred_covs, np.float64) * cov_scale), _VUHW_TO_CORNERS.T)
        boxes = vuhw_to_vuvu_np(np.asarray(pred_means_vuhw))
        if class_columns is not None:
            pred_cat_params = np.stack([np.asarray(pred_cat_params)[:, c] for c in class_columns], axis=1)
        for cat, b, cv in zip(pred_cat_params, boxes, covs):
            if np.max(cat) >= score_threshold:
                dets.append(PBoxDetInst(cat, np.array([b[1], b[0], b[3], b[2]]).astype(np.int32), [cv[0:2, 0:2], cv[2:4, 2:4]]))
    return gts, dets


def evaluate(matches):
    """PDQ of a list of (gt_instances, det_instances): the row the drivers print (score in percent)."""
    ev = PDQ()
    score = ev.score(matches) * 100
    tp, fp, fn = ev.get_assignment_counts()
    return {'score': float(score), 'TP': int(tp), 'FP': int(fp), 'FN': int(fn), 'avg_spatial_quality': float(ev.get_avg_spatial_score()),
            'avg_label_quality': float(ev.get_avg_label_score()), 'avg_overall_quality': float(ev.get_avg_overall_quality_score())}
