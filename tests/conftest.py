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


# Order of the GPU suite (round 6): the driver runs `pytest -x -m gpu` against a clock, so the ORACLE-PARITY files come first -- a late
# failure or a kill at the limit must never hide them -- then the bit-identity / property suites, then the timing sweeps and the
# multi-process tests, then the tests on pipeline_overlap handles (which may report DESIGN.md 8.4's known issue), and the 8.4 canary LAST.
_GPU_FILE_ORDER = ["test_gpu_baseline_sizes.py", "test_gpu_post.py", "test_gpu_conv.py", "test_gpu_loss.py", "test_gpu_preprocess.py",
                   "test_gpu_production_kernel.py", "test_gpu_forward.py", "test_gpu_pipeline.py", "test_gpu_train_blocks.py",
                   "test_gpu_train_step.py", "test_gpu_markers.py", "test_gpu_planner.py"]
_GPU_LATE = ("two_ranks", "dead_peer", "under_rccl", "test_default_plan_is_never_beaten")      # sweeps and subprocess tests: after the parity files
_GPU_LAST = ("overlapped_pipeline", "canary")                                                 # overlap handles, then the canary


def _gpu_rank(item):
    if "gpu" not in item.keywords:
        return (0, 0)
    name = item.name
    if "canary" in name:
        return (4, 0)
    if any(k in name for k in _GPU_LAST):
        return (3, 0)
    if any(k in name for k in _GPU_LATE):
        return (2, 0)
    fname = os.path.basename(str(item.fspath))
    return (1, _GPU_FILE_ORDER.index(fname) if fname in _GPU_FILE_ORDER else len(_GPU_FILE_ORDER))


def pytest_collection_modifyitems(config, items):
    items.sort(key=_gpu_rank)            # (stable: the order inside a file stays)
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


def compare_posterior(got, ref, uniforms, tol=1e-3, mean_floor=1.0, cov_tol=None, exact_counts=True, min_checked=20,
                      max_ambiguous=2e-2, boundary_eps=1e-5):
    """Posterior of the device (`got` = Engine.get_posterior) against the oracle's (`ref` =
    bayes_od_posterior(..., return_debug=True)) WITHOUT an all-or-nothing guard on the kept set.

    Anchors whose categorical draw lies within 1e-5 of a CDF boundary may legitimately sample the neighbouring class on
    the device (fp32 cumsum vs the oracle's dtype; `boundary_eps` is raised when the class probabilities themselves come
    from a different forward pass, e.g. the end-to-end tests); they are flagged `ambiguous` and excluded.  Every other anchor
    must agree on the background filter, and the anchors both sides keep are compared element-wise.  Returns the
    number of anchors compared (asserted >= min_checked, so the test can never pass vacuously).  The expected ambiguous
    rate is draws x classes x 2 boundary_eps = 30 x 8 x 2e-5 = 0.5 % of the anchors at the default."""
    a = ref["keep"].shape[0]
    cdf = np.cumsum(ref["mean_probs"], axis=1)
    t = uniforms.astype(np.float64) * cdf[:, -1:]
    ambiguous = np.abs(cdf[:, None, :] - t[:, :, None]).min(axis=(1, 2)) < boundary_eps
    ref_keep = ref["keep"]
    got_keep = np.zeros(a, bool)
    got_keep[got["anchor_index"]] = True
    assert np.all(got["anchor_index"][1:] > got["anchor_index"][:-1])           # tf.boolean_mask order
    assert not np.any((got_keep != ref_keep) & ~ambiguous), "filter mismatch on an unambiguous anchor"
    assert ambiguous.mean() < max_ambiguous
    both = got_keep & ref_keep & ~ambiguous
    gi = np.searchsorted(got["anchor_index"], np.nonzero(both)[0])
    ri = np.cumsum(ref_keep)[both] - 1
    assert len(gi) >= min_checked, "only %d anchors compared" % len(gi)
    if exact_counts:
        assert np.array_equal(got["counts"][gi], ref["counts"][ri].astype(np.float32))
    assert rel_err(got["score"][gi], ref["score"][ri], 1e-6) < tol
    assert rel_err(got["means"][gi], ref["means"][ri][:, :, 0], mean_floor) < tol
    cov_ref = ref["covs"][ri]
    floor = np.abs(cov_ref).reshape(len(ri), -1).max(axis=1)[:, None, None] * 1e-2
    err = np.abs(got["covs"][gi] - cov_ref) / (np.abs(cov_ref) + floor)
    assert err.max() < (tol if cov_tol is None else cov_tol), float(err.max())
    return len(gi), bool(np.array_equal(got_keep, ref_keep))
