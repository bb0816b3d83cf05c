import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


ANCHOR_CFG = {"layers": [3, 4, 5, 6, 7],
              "aspect_ratios": [[1.0, 1.0], [1.0, 2.0], [2.0, 1.0]],
              "scales": [1.0, 1.26, 1.59], "min_positive_iou": 0.5, "max_negative_iou": 0.4}
BAYES_CFG = {"ranking_method": "score", "dirichlet_prior": {"type": "non_informative"},
             "gaussian_prior": {"type": "isotropic", "isotropic_variance": 100000.0},
             "fusion_method": "none"}
NMS_CFG = {"max_output_size": 100, "iou_threshold": 0.5, "soft_nms_sigma": 0.5}


def rel_err(a, b, floor):
    """max |a-b| / (|b| + floor)"""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b) / (np.abs(b) + floor)))
