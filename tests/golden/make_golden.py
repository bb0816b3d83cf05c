#!/usr/bin/env python3
"""Capture golden input/output vectors from the reference's importable NumPy half.

Run ONCE in the build container (``/root/reference`` must exist); the GPU box never
runs this.  TensorFlow / tensorflow-probability are not installed, so the module-level
``import tensorflow`` in the reference files is satisfied by an empty stub whose only
attribute is an identity ``tf.function`` decorator.  Only NumPy/SciPy code
of the reference is executed:

  * ``bayes_od_clustering``      src/retina_net/experiments/inference_utils.py:285-364
  * ``map_dataset_classes``      src/retina_net/experiments/inference_utils.py:372-404
  * ``vuhw_to_vuvu_np`` / ``vuvu_to_vuhw_np``   src/retina_net/anchor_generator/box_utils.py:49-91
  * ``compute_gaussian_entropy_np`` / ``compute_categorical_entropy_np``
                                 src/core/evaluation_utils_2d.py:280-290
  * ``two_d_iou`` / ``get_ap``   src/core/evaluation_utils_2d.py:12-49,253-269
  * ``predictions_to_bdd_format`` / ``predictions_to_kitti_format`` / ``strip_checkpoint_id``
                                 src/retina_net/experiments/validation_utils.py:96-107,183-272

  * PDQ: ``PBoxDetInst`` / ``BBoxDetInst`` heatmaps, ``gen_single_heatmap``, ``_calc_qual_img``, the ``PDQ`` totals
                                 src/retina_net/offline_eval/pdq_data_holders.py:81-268, pdq.py:11-452
                                 (``make_golden.py pdq`` writes only ``pdq.npz``)

Outputs (data only, no reference source text): ``tests/golden/*.npz`` + ``writers.json``.
"""
import json
import os
import sys
import types

import numpy as np

REF = os.environ.get("BAYESOD_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_tf_stub():
    tf = types.ModuleType("tensorflow")
    tf.function = lambda f=None, **kw: f if f is not None else (lambda g: g)
    tf.keras = types.SimpleNamespace()
    tfp = types.ModuleType("tensorflow_probability")
    sys.modules["tensorflow"] = tf
    sys.modules["tensorflow_probability"] = tfp
    # numpy aliases removed in numpy>=1.24 that a few reference helpers still use
    for name, typ in (("int", int), ("float", float), ("bool", bool)):
        if not hasattr(np, name):
            setattr(np, name, typ)


def _random_spd(rng, n, scale):
    a = rng.normal(size=(n, 4, 4))
    m = a @ np.transpose(a, (0, 2, 1)) + 0.5 * np.eye(4)[None]
    return (m * scale).astype(np.float32)


def _boxes_vuhw(rng, n, n_objects):
    """n boxes jittered around n_objects centres so IoU clusters exist."""
    centres = rng.uniform(60, 440, size=(n_objects, 2))
    dims = rng.uniform(30, 120, size=(n_objects, 2))
    which = rng.integers(0, n_objects, size=n)
    vu = centres[which] + rng.normal(scale=2.0, size=(n, 2))
    hw = dims[which] * np.exp(rng.normal(scale=0.04, size=(n, 2)))
    return np.concatenate([vu, hw], axis=1).astype(np.float32), which


def _iou_matrix_ref_convention(vuvu):
    """IoU exactly as box_utils.bbox_iou_vuvu:117-146 (float32, +1 convention, area quirk)."""
    b = vuvu.astype(np.float32)
    y1, x1, y2, x2 = b[:, 0:1], b[:, 1:2], b[:, 2:3], b[:, 3:4]
    xi1 = np.maximum(x1, x1.T)
    yi1 = np.maximum(y1, y1.T)
    xi2 = np.minimum(x2, x2.T)
    yi2 = np.minimum(y2, y2.T)
    one = np.float32(1.0)
    inter = np.maximum(xi2 - xi1 + one, 0) * np.maximum(yi2 - yi1 + one, 0)
    area = (x1 - x2 + one) * (y1 - y2 + one)
    union = area + area.T - inter
    return (inter / (union + np.float32(0.00001))).astype(np.float32)


def golden_pdq():
    """Inputs and the reference's outputs for the PDQ metric (small 40x56 images so the file stays small)."""
    from src.retina_net.offline_eval import pdq as ref_pdq, pdq_data_holders as ref_dh
    rng = np.random.default_rng(2024)
    shape = (40, 56)
    out = {"img_shape": np.array(shape)}
    # ---- Gaussian corner heatmaps: interior, touching the top / left edges (outside-mass correction), near-singular
    means = np.array([[20.3, 30.7], [1.2, 25.0], [18.0, 0.6], [0.4, 0.9], [35.5, 50.2], [12.0, 12.0]])
    covs = np.array([[[4.0, 0.8], [0.8, 6.0]], [[3.0, -0.5], [-0.5, 2.0]], [[5.0, 1.0], [1.0, 1.5]],
                     [[2.0, 0.3], [0.3, 2.5]], [[9.0, -2.0], [-2.0, 7.0]], [[1e-6, 0.0], [0.0, 1e-6]]])
    out["corner_means"] = means
    out["corner_covs"] = covs
    out["corner_heatmaps"] = np.stack([ref_dh.gen_single_heatmap(shape, list(m), c) for m, c in zip(means, covs)])
    out["corner_rois"] = np.array([list(ref_dh.find_roi(shape, list(m), c)) for m, c in zip(means, covs)])
    # ---- probabilistic boxes and a plain fractional box
    boxes = np.array([[10, 8, 30, 25], [2, 1, 20, 12], [35, 20, 54, 38], [5, 5, 50, 35]])
    bcovs = np.stack([np.stack([_random_spd(rng, 1, 1.0)[0][:2, :2], _random_spd(rng, 1, 2.0)[0][:2, :2]]) for _ in boxes]).astype(np.float64)
    probs = rng.dirichlet(np.ones(5) * 0.5, size=len(boxes))
    out["pbox_boxes"], out["pbox_covs"], out["pbox_probs"] = boxes, bcovs, probs
    out["pbox_heatmaps"] = np.stack([ref_dh.PBoxDetInst(p, b, [c[0], c[1]]).calc_heatmap(shape) for p, b, c in zip(probs, boxes, bcovs)])
    fbox = np.array([10.3, 7.6, 30.2, 21.9])
    out["bbox_box"] = fbox
    out["bbox_heatmap"] = ref_dh.BBoxDetInst(probs[0], fbox, 0.8).calc_heatmap(shape)

    # ---- whole images: lists of ground-truth boxes / detections -> _calc_qual_img sums, then the PDQ totals
    def image(gt_boxes, gt_labels, det_idx):
        gts = []
        for b, l in zip(gt_boxes, gt_labels):
            m = np.zeros(shape, dtype=bool)
            m[b[1]:b[3], b[0]:b[2]] = True
            gts.append(ref_dh.GroundTruthInstance(m, int(l), 0, 0, bounding_box=np.array(b)))
        dets = [ref_dh.PBoxDetInst(probs[i], boxes[i], [bcovs[i][0], bcovs[i][1]]) for i in det_idx]
        return gts, dets
    images = [
        ([[10, 8, 30, 25], [36, 21, 53, 37], [1, 30, 9, 38]], [int(np.argmax(probs[0])), 2, 1], [0, 2, 1, 3]),   # last gt too small
        ([[4, 4, 49, 34], [2, 1, 19, 12]], [int(np.argmax(probs[3])), int(np.argmax(probs[1]))], [3]),
        ([], [], [0, 1]),
        ([[10, 8, 30, 25], [0, 0, 5, 5]], [0, 1], []),
        ([[20, 10, 40, 30]], [4], [1]),                                                                        # disjoint: FP + FN
    ]
    ev = ref_pdq.PDQ()
    res = []
    for k, (gb, gl, di) in enumerate(images):
        gts, dets = image(gb, gl, di)
        r = ref_pdq._calc_qual_img(gts, dets)
        res.append([float(r['overall']), float(r['spatial']), float(r['label']), r['TP'], r['FP'], r['FN']])
        ev.add_img_eval(gts, dets)
        out["img%d_gt_boxes" % k] = np.array(gb, dtype=np.int64).reshape(-1, 4)
        out["img%d_gt_labels" % k] = np.array(gl, dtype=np.int64)
        out["img%d_det_idx" % k] = np.array(di, dtype=np.int64)
    out["n_images"] = np.array(len(images))
    out["image_results"] = np.array(res, dtype=np.float64)
    out["pdq_totals"] = np.array([ev.get_pdq_score(), ev.get_avg_spatial_score(), ev.get_avg_label_score(),
                                  ev.get_avg_overall_quality_score()] + list(ev.get_assignment_counts()), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "pdq.npz"), **out)
    print("pdq.npz written")


def golden_affinity():
    """bayes_od_clustering with a caller-supplied affinity that is NOT the IoU of the means (inference_utils.py:290,316):
    a Gaussian kernel on the box-centre distance, threshold 0.6.  Writes only ``clustering_affinity.npz``."""
    from src.retina_net.experiments import inference_utils as ref_iu
    cases, idx = {}, 0
    for seed in range(3):
        for (m, n_obj, k, c) in ((5, 2, 2, 8), (40, 4, 6, 8), (150, 9, 20, 4)):
            rng = np.random.default_rng(7000 + 100 * seed + m)
            vuhw, _ = _boxes_vuhw(rng, m, n_obj)
            d2 = ((vuhw[:, None, :2] - vuhw[None, :, :2]) ** 2).sum(-1)
            aff = np.exp(-d2 / np.float32(2.0 * 6.0 ** 2)).astype(np.float32)
            aff[np.abs(aff - 0.6) < 1e-3] = 0.7          # keep every entry away from the threshold
            covs = _random_spd(rng, m, 4.0)
            probs = rng.dirichlet(np.ones(c) * 0.6, size=m)
            counts = np.stack([rng.multinomial(30, p) for p in probs]).astype(np.float32) + np.float32(1.0 / c)
            centres = rng.choice(m, size=min(k, m), replace=False).astype(np.int32)
            means = vuhw[:, :, None].astype(np.float32)
            scores, fmeans, fcovs, fcounts = ref_iu.bayes_od_clustering(counts, means, covs, centres, aff, affinity_threshold=0.6)
            tag = "a%02d" % idx
            for name, val in (("counts", counts), ("means", means), ("covs", covs), ("centres", centres), ("affinity", aff),
                              ("out_scores", scores), ("out_means", fmeans), ("out_covs", fcovs), ("out_counts", fcounts)):
                cases["%s_%s" % (tag, name)] = np.asarray(val)
            idx += 1
    cases["n_cases"] = np.int32(idx)
    cases["affinity_threshold"] = np.float32(0.6)
    np.savez_compressed(os.path.join(OUT, "clustering_affinity.npz"), **cases)
    print("clustering_affinity.npz written (%d cases)" % idx)


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not found at %s" % REF)
    _install_tf_stub()
    sys.path.insert(0, REF)
    if len(sys.argv) > 1 and sys.argv[1] == "pdq":
        golden_pdq()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "affinity":
        golden_affinity()
        return
    from src.retina_net.experiments import inference_utils as ref_iu
    from src.retina_net.anchor_generator import box_utils as ref_bu
    from src.core import evaluation_utils_2d as ref_ev
    from src.retina_net.experiments import validation_utils as ref_vu

    # ------------------------------------------------------------------ clustering
    cases = {}
    idx = 0
    for seed in range(5):
        for (m, n_obj, k, c) in ((1, 1, 1, 8), (4, 1, 1, 8), (12, 3, 2, 4),
                                 (60, 5, 5, 8), (200, 12, 17, 8)):
            rng = np.random.default_rng(1000 * seed + m)
            vuhw, which = _boxes_vuhw(rng, m, n_obj)
            vuvu = ref_bu.vuhw_to_vuvu_np(vuhw)
            iou = _iou_matrix_ref_convention(vuvu)
            covs = _random_spd(rng, m, 4.0)
            # counts: 30 categorical draws + 1/C prior, with distinct rows to avoid KL ties
            probs = rng.dirichlet(np.ones(c) * 0.6, size=m)
            counts = np.stack([rng.multinomial(30, p) for p in probs]).astype(np.float32)
            counts = counts + np.float32(1.0 / c)
            k_eff = min(k, m)
            centres = rng.choice(m, size=k_eff, replace=False).astype(np.int32)
            means = vuhw[:, :, None].astype(np.float32)
            scores, fmeans, fcovs, fcounts = ref_iu.bayes_od_clustering(
                counts, means, covs, centres, iou, affinity_threshold=0.5)
            # record KL margins so the test can skip argpartition ties
            tag = "c%02d" % idx
            cases[tag + "_counts"] = counts
            cases[tag + "_means"] = means
            cases[tag + "_covs"] = covs
            cases[tag + "_centres"] = centres
            cases[tag + "_iou"] = iou
            cases[tag + "_out_scores"] = np.asarray(scores)
            cases[tag + "_out_means"] = np.asarray(fmeans)
            cases[tag + "_out_covs"] = np.asarray(fcovs)
            cases[tag + "_out_counts"] = np.asarray(fcounts)
            idx += 1
    cases["n_cases"] = np.int32(idx)
    cases["affinity_threshold"] = np.float32(0.5)
    np.savez_compressed(os.path.join(OUT, "clustering.npz"), **cases)

    # ------------------------------------------------------------------ box utils
    rng = np.random.default_rng(7)
    vuhw = np.concatenate([rng.uniform(0, 500, (64, 2)), rng.uniform(1, 200, (64, 2))], 1)
    vuhw32 = vuhw.astype(np.float32)
    np.savez_compressed(
        os.path.join(OUT, "box_utils.npz"),
        vuhw64=vuhw, vuvu64=ref_bu.vuhw_to_vuvu_np(vuhw),
        back64=ref_bu.vuvu_to_vuhw_np(ref_bu.vuhw_to_vuvu_np(vuhw)),
        vuhw32=vuhw32, vuvu32=ref_bu.vuhw_to_vuvu_np(vuhw32),
        back32=ref_bu.vuvu_to_vuhw_np(ref_bu.vuhw_to_vuvu_np(vuhw32)))

    # ------------------------------------------------------------------ class mapping
    rng = np.random.default_rng(11)
    sc = rng.dirichlet(np.ones(8), size=40).astype(np.float32)
    mapped = ref_iu.map_dataset_classes("bdd", "kitti", sc)
    same = ref_iu.map_dataset_classes("coco", "pascal", sc)  # empty mapping dict -> identity
    np.savez_compressed(os.path.join(OUT, "class_map.npz"),
                        scores=sc, bdd_to_kitti=mapped, identity=same)

    # ------------------------------------------------------------------ entropies / eval helpers
    rng = np.random.default_rng(13)
    covs = _random_spd(rng, 16, 3.0).astype(np.float64)
    g_ent = np.array([ref_ev.compute_gaussian_entropy_np(c) for c in covs])
    cat = rng.dirichlet(np.ones(8), size=16)
    c_ent = np.array([ref_ev.compute_categorical_entropy_np(c) for c in cat])
    box = np.array([10., 20., 110., 220.])
    boxes = np.concatenate([rng.uniform(0, 100, (32, 2)), rng.uniform(100, 300, (32, 2))], 1)
    iou = ref_ev.two_d_iou(box, boxes)
    rec = np.sort(rng.uniform(0, 1, 50))
    prec = np.sort(rng.uniform(0, 1, 50))[::-1].copy()
    ap = ref_ev.get_ap(rec.copy(), prec.copy())
    np.savez_compressed(os.path.join(OUT, "eval_helpers.npz"),
                        covs=covs, gaussian_entropy=g_ent, cat=cat, categorical_entropy=c_ent,
                        box=box, boxes=boxes, two_d_iou=iou, recalls=rec, precisions=prec,
                        ap=np.float64(ap))

    # ------------------------------------------------------------------ AP / minimum uncertainty error (f4)
    def _records(rng, n_img, n_gt, n_pred):
        cats = ["car", "person", "truck"]
        gt, pred = [], []
        for i in range(n_img):
            for _ in range(int(rng.integers(0, n_gt + 1))):
                x, y = rng.uniform(0, 300, 2)
                w, h = rng.uniform(20, 120, 2)
                gt.append({"name": "im%03d" % i, "category": cats[int(rng.integers(0, 3))],
                           "bbox": [float(x), float(y), float(x + w), float(y + h)]})
        for g in gt:                                   # detections: jittered copies of the ground truth + clutter
            if rng.uniform() < 0.8:
                j = rng.normal(0, 6, 4)
                pred.append({"name": g["name"], "category": g["category"] if rng.uniform() < 0.9 else cats[int(rng.integers(0, 3))],
                             "bbox": [float(v) for v in (np.asarray(g["bbox"]) + j)],
                             "score": float(rng.uniform(0.3, 1.0)), "entropy_score": float(rng.uniform(0, 3))})
        for _ in range(n_pred):
            x, y = rng.uniform(0, 300, 2)
            w, h = rng.uniform(20, 120, 2)
            pred.append({"name": "im%03d" % int(rng.integers(0, n_img + 2)), "category": cats[int(rng.integers(0, 3))],
                         "bbox": [float(x), float(y), float(x + w), float(y + h)],
                         "score": float(rng.uniform(0.0, 0.7)), "entropy_score": float(rng.uniform(1, 5))})
        return gt, pred
    import copy
    eval_cases = []
    for seed, (n_img, n_gt, n_pred) in enumerate(((4, 3, 5), (12, 5, 30), (30, 4, 80))):
        gt, pred = _records(np.random.default_rng(100 + seed), n_img, n_gt, n_pred)
        thr = [0.5] if seed != 1 else [0.7]     # one threshold per call, as every script of the reference does
        # (cat_pc indexes the prediction list with a FLAT arg-max over [detections, thresholds]: more than one
        # threshold overruns the list, evaluation_utils_2d.py:123)
        m_ap, aps, cats, opt, fmax = ref_ev.evaluate_detection(copy.deepcopy(gt), copy.deepcopy(pred), thr)
        mues, mue, cats_u, at = ref_ev.evaluate_u_error(copy.deepcopy(gt), copy.deepcopy(pred), thr)
        eval_cases.append({"gt": gt, "pred": pred, "thresholds": thr, "mAP": float(m_ap), "aps": aps, "cat_list": cats,
                           "optimal_score_thresholds": opt, "maximum_f_scores": fmax, "min_u_errors": mues,
                           "min_u_error": float(mue), "scores_at_min_u_errors": at})
    with open(os.path.join(OUT, "eval_metrics.json"), "w") as fp:
        json.dump(eval_cases, fp)

    # ------------------------------------------------------------------ writers
    rng = np.random.default_rng(17)
    out_boxes = np.concatenate([rng.uniform(0, 200, (6, 2)), rng.uniform(200, 500, (6, 2))], 1)
    out_boxes = out_boxes.astype(np.float32)
    cls8 = rng.dirichlet(np.ones(8) * 0.3, size=6).astype(np.float32)
    cls8[0] = np.eye(8, dtype=np.float32)[7] * 0.9 + 0.0125   # background-dominant row is dropped
    bdd = ref_vu.predictions_to_bdd_format(
        out_boxes, cls8, "frame_0001",
        category_list=['car', 'truck', 'bus', 'person', 'rider', 'bike', 'motor'])
    cls5 = ref_iu.map_dataset_classes("bdd", "kitti", cls8)
    kitti = ref_vu.predictions_to_kitti_format(out_boxes, cls5)
    with open(os.path.join(OUT, "writers.json"), "w") as fp:
        json.dump({
            "boxes": out_boxes.tolist(), "cls8": cls8.tolist(), "cls5": cls5.tolist(),
            "bdd": bdd, "kitti": [[str(v) for v in row] for row in kitti.tolist()],
            "ckpt_ids": {p: ref_vu.strip_checkpoint_id(p)
                         for p in ("a/b/ckpt-101", "x/retinanet_bdd_covar-7", "ckpt-000012")},
        }, fp, indent=1)
    golden_pdq()
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
