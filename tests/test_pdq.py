"""CPU: the PDQ restatement (bayes_od_rc_amd/prob_detection_quality.py) against vectors
captured from the reference's own functions (tests/golden/pdq.npz, generator tests/golden/make_golden.py pdq):
Gaussian-corner heatmaps and their regions of interest, probabilistic-box and plain-box heatmaps, per-image quality
sums (Hungarian assignment, small-object rule, false-positive credit) and the accumulated PDQ totals.
Tolerance 1e-6 absolute on probabilities / qualities (float32 heatmaps), counts exact."""
import os

import numpy as np
import pytest

from bayes_od_rc_amd import prob_detection_quality as pdq

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pdq.npz"))
SHAPE = tuple(int(v) for v in G["img_shape"])


def test_corner_regions_and_heatmaps():
    for k, (m, c) in enumerate(zip(G["corner_means"], G["corner_covs"])):
        assert list(pdq.corner_roi(SHAPE, list(m), c)) == list(G["corner_rois"][k]), k
        np.testing.assert_allclose(pdq.corner_heatmap(SHAPE, list(m), c), G["corner_heatmaps"][k], atol=1e-6, rtol=0)


def test_box_heatmaps():
    for p, b, c, want in zip(G["pbox_probs"], G["pbox_boxes"], G["pbox_covs"], G["pbox_heatmaps"]):
        got = pdq.PBoxDetInst(p, b, [c[0], c[1]]).calc_heatmap(SHAPE)
        np.testing.assert_allclose(got, want, atol=1e-6, rtol=0)
        assert got.max() <= 1 and (got[(got > 0)] >= pdq.HEATMAP_FLOOR).all()
    got = pdq.BBoxDetInst(G["pbox_probs"][0], G["bbox_box"], 0.8).calc_heatmap(SHAPE)
    np.testing.assert_allclose(got, G["bbox_heatmap"], atol=1e-7, rtol=0)


def _image(k):
    gts = []
    for b, l in zip(G["img%d_gt_boxes" % k], G["img%d_gt_labels" % k]):
        m = np.zeros(SHAPE, dtype=bool)
        m[b[1]:b[3], b[0]:b[2]] = True
        gts.append(pdq.GroundTruthInstance(m, int(l), 0, 0, bounding_box=np.array(b)))
    dets = [pdq.PBoxDetInst(G["pbox_probs"][i], G["pbox_boxes"][i], [G["pbox_covs"][i][0], G["pbox_covs"][i][1]])
            for i in G["img%d_det_idx" % k]]
    return gts, dets


def test_image_sums_and_totals():
    ev = pdq.PDQ()
    for k in range(int(G["n_images"])):
        gts, dets = _image(k)
        r = pdq.image_quality(gts, dets)
        want = G["image_results"][k]
        np.testing.assert_allclose([r["overall"], r["spatial"], r["label"]], want[:3], atol=2e-6, rtol=0, err_msg=str(k))
        assert [r["TP"], r["FP"], r["FN"]] == [int(v) for v in want[3:]], k
        ev.add_img_eval(gts, dets)
    tot = G["pdq_totals"]
    np.testing.assert_allclose([ev.get_pdq_score(), ev.get_avg_spatial_score(), ev.get_avg_label_score(),
                                ev.get_avg_overall_quality_score()], tot[:4], atol=2e-6, rtol=0)
    assert list(ev.get_assignment_counts()) == [int(v) for v in tot[4:]]
    # score() over the list of images gives the same total (the reference fans out over a process pool)
    assert abs(pdq.PDQ().score([_image(k) for k in range(int(G["n_images"]))]) - tot[0]) < 2e-6


def test_known_answers():
    """A detection that reproduces its object with certainty scores 1; an empty image contributes nothing."""
    m = np.zeros(SHAPE, dtype=bool)
    m[8:25, 10:30] = True
    gt = pdq.GroundTruthInstance(m, 2, 0, 0)
    assert gt.bounding_box == [10, 8, 29, 24] and gt.num_pixels == 17 * 20 and pdq.gt_counts_for_pdq(gt)
    perfect = pdq.DetectionInstance(np.eye(5)[2], heatmap=m.astype(np.float32))
    r = pdq.image_quality([gt], [perfect])
    assert r["TP"] == 1 and r["FP"] == 0 and r["FN"] == 0 and abs(r["overall"] - 1.0) < 1e-6
    r = pdq.image_quality([], [])
    assert r == {'overall': 0.0, 'spatial': 0.0, 'label': 0.0, 'TP': 0, 'FP': 0, 'FN': 0}
    with pytest.raises(ValueError):
        pdq.mask_bounding_box(np.zeros((4, 4), dtype=bool))


def test_frame_instances_from_prediction_files():
    """The compute_pdq drivers' conversion: vuhw mean / 4x4 covariance / class parameters of one frame -> instances."""
    means = np.array([[16.0, 20.0, 16.0, 20.0], [30.0, 40.0, 8.0, 10.0]])            # v, u, h, w
    covs = np.stack([np.diag([0.02, 0.03, 0.04, 0.05]), np.diag([0.01, 0.01, 0.01, 0.01])])
    cats = np.array([[0.9, 0.05, 0.05], [0.3, 0.4, 0.3]])                            # the second is below the 0.5445 threshold
    gts, dets = pdq.frame_instances(np.array([[0, 1, 0]]), np.array([[10.0, 8.0, 30.0, 24.0]]), means, covs, cats, SHAPE)
    assert len(gts) == 1 and gts[0].class_label == 1 and len(dets) == 1
    assert list(dets[0].box) == [10, 8, 30, 24]
    # corner covariance = T cov T^T * 70: var(x1) = (var_u + var_w / 4) * 70
    np.testing.assert_allclose(dets[0].covs[0][0, 0], (0.03 + 0.05 / 4) * 70, rtol=1e-12)
    np.testing.assert_allclose(dets[0].covs[1][1, 1], (0.02 + 0.04 / 4) * 70, rtol=1e-12)
    out = pdq.evaluate([(gts, dets)])
    assert 0.0 <= out["score"] <= 100.0 and out["TP"] + out["FP"] + out["FN"] >= 1


def test_kitti_driver_options():
    """kitti/compute_pdq.py:66-118: v-u ordered label boxes, car / person columns of the BDD-trained class vector,
    0.5 threshold, clipping to the 1300-wide canvas."""
    means = np.array([[16.0, 20.0, 16.0, 20.0], [30.0, 40.0, 8.0, 10.0]])
    covs = np.stack([np.diag([0.02, 0.03, 0.04, 0.05]), np.diag([0.01, 0.01, 0.01, 0.01])])
    cats = np.array([[0.52, 0.2, 0.1, 0.1, 0.02, 0.02, 0.02, 0.02], [0.1, 0.6, 0.1, 0.1, 0.025, 0.025, 0.025, 0.025]])
    gts, dets = pdq.frame_instances(np.array([[0, 1, 0, 0]]), np.array([[8.0, 10.0, 24.0, 30.0]]), means, covs, cats, SHAPE,
                                    score_threshold=0.5, class_columns=(0, 3), gt_boxes_vuvu=True, clip_max=1300)
    assert list(gts[0].bounding_box) == [10, 8, 30, 24] and gts[0].class_label == 1
    assert len(dets) == 1 and list(dets[0].class_list) == [0.52, 0.1]                   # the truck-dominant detection is dropped
    bdd_gts, bdd_dets = pdq.frame_instances(np.array([[0, 1, 0]]), np.array([[10.0, 8.0, 30.0, 24.0]]), means, covs, cats[:, :3], SHAPE,
                                            score_threshold=0.5)
    np.testing.assert_array_equal(gts[0].segmentation_mask, bdd_gts[0].segmentation_mask)
    np.testing.assert_allclose(dets[0].covs[0], bdd_dets[0].covs[0], rtol=1e-12)
