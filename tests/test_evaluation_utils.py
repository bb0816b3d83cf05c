"""CPU: offline metrics (bayes_od_rc_amd/evaluation_utils_2d.py) against values captured from the reference's
src/core/evaluation_utils_2d.py by import (tests/golden/make_golden.py -> eval_metrics.json, eval_helpers.npz)."""
import copy
import json
import os

import numpy as np


def test_ap_and_uncertainty_error_match_the_reference(golden_dir):
    from bayes_od_rc_amd import evaluation_utils_2d as ev
    cases = json.load(open(os.path.join(golden_dir, "eval_metrics.json")))
    assert len(cases) == 3
    for c in cases:
        m_ap, aps, cats, opt, fmax = ev.evaluate_detection(copy.deepcopy(c["gt"]), copy.deepcopy(c["pred"]), c["thresholds"])
        assert cats == c["cat_list"]
        assert abs(m_ap - c["mAP"]) < 1e-9 and np.allclose(aps, c["aps"], atol=1e-9)
        assert np.allclose(opt, c["optimal_score_thresholds"], atol=0) and np.allclose(fmax, c["maximum_f_scores"], atol=1e-12)
        mues, mue, cats_u, at = ev.evaluate_u_error(copy.deepcopy(c["gt"]), copy.deepcopy(c["pred"]), c["thresholds"])
        assert cats_u == c["cat_list"] and abs(mue - c["min_u_error"]) < 1e-12
        assert np.allclose(mues, c["min_u_errors"], atol=1e-12) and np.allclose(at, c["scores_at_min_u_errors"], atol=0)
    assert cases[2]["mAP"] > 10.0                      # the synthetic detections do match their ground truth


def test_helpers_match_the_reference(golden_dir):
    from bayes_od_rc_amd import evaluation_utils_2d as ev
    g = np.load(os.path.join(golden_dir, "eval_helpers.npz"))
    assert np.array_equal(ev.two_d_iou(g["box"], g["boxes"]), g["two_d_iou"])
    assert abs(ev.get_ap(g["recalls"].copy(), g["precisions"].copy()) - float(g["ap"])) < 1e-12
    assert np.allclose([ev.compute_gaussian_entropy_np(c) for c in g["covs"]], g["gaussian_entropy"], rtol=1e-12)
    assert np.allclose([ev.compute_categorical_entropy_np(c) for c in g["cat"]], g["categorical_entropy"], rtol=1e-12)


def test_edge_cases():
    from bayes_od_rc_amd import evaluation_utils_2d as ev
    gt = [{"name": "a", "category": "car", "bbox": [0, 0, 10, 10]}]
    # a category with no predictions contributes AP 0; predictions on images without ground truth are false positives
    m_ap, aps, cats, _, _ = ev.evaluate_detection(gt, [{"name": "zz", "category": "bus", "bbox": [0, 0, 1, 1], "score": 0.5}])
    assert cats == ["car"] and aps == [0.0] and m_ap == 0.0
    pred = [{"name": "a", "category": "car", "bbox": [0, 0, 10, 10], "score": 0.9},
            {"name": "a", "category": "car", "bbox": [0, 0, 10, 10], "score": 0.8},      # duplicate: false positive
            {"name": "b", "category": "car", "bbox": [0, 0, 10, 10], "score": 0.7}]
    m_ap, aps, _, opt, fmax = ev.evaluate_detection(gt, pred)
    assert aps == [100.0] and opt == [0.9] and abs(fmax[0] - 1.0) < 1e-5
