"""Known answers of SURVEY.md App. A.9 for anchors / decode / IoU / targets (oracle and host mirror)."""
import numpy as np
import pytest

from conftest import ANCHOR_CFG


def _gens():
    from oracle import geometry
    from bayes_od_rc_amd.anchor_generator import FpnAnchorGenerator
    g = FpnAnchorGenerator(ANCHOR_CFG)
    return [lambda shape, l: geometry.generate_anchors(shape, l, ANCHOR_CFG["aspect_ratios"], ANCHOR_CFG["scales"]),
            g.generate_anchors]


@pytest.mark.parametrize("shape,counts", [((512, 512, 3), [4096, 1024, 256, 64, 16]),
                                          ((384, 1248, 3), [48 * 156, 24 * 78, 12 * 39, 6 * 20, 3 * 10]),
                                          ((720, 1280, 3), [90 * 160, 45 * 80, 23 * 40, 12 * 20, 6 * 10])])
def test_anchor_counts(shape, counts):
    for gen in _gens():
        per_level = [gen(shape, l).shape[0] for l in ANCHOR_CFG["layers"]]
        assert per_level == [9 * c for c in counts]


def test_anchor_known_values_512():
    a_or, a_host = [g((512, 512, 3), 3) for g in _gens()]
    assert np.array_equal(a_or, a_host) and a_or.dtype == np.float32
    a = a_or
    assert a.shape == (36864, 4)
    assert tuple(a[0, :2]) == (4.0, 4.0)                        # first centre (i+0.5)*stride
    assert np.allclose(a[0:3, 2], 32 * np.array([1.0, 1.26, 1.59]), rtol=1e-6)   # ratio [1,1] x scales
    assert np.allclose(a[0:3, 3], 32 * np.array([1.0, 1.26, 1.59]), rtol=1e-6)
    assert np.allclose(a[3, 2:], [32 / np.sqrt(2), 32 * np.sqrt(2)], rtol=1e-6)   # ratio [1,2]
    assert np.allclose(a[6, 2:], [32 * np.sqrt(2), 32 / np.sqrt(2)], rtol=1e-6)   # ratio [2,1]
    assert tuple(a[9, :2]) == (4.0, 12.0)                       # u is the inner loop
    assert tuple(a[9 * 64, :2]) == (12.0, 4.0)
    top = _gens()[0]((512, 512, 3), 7)
    assert top.shape == (144, 4) and tuple(top[0, :2]) == (64.0, 64.0) and top[0, 2] == 512.0


def test_decode_zero_target_returns_anchor_and_clip():
    from oracle import geometry
    from bayes_od_rc_amd import box_utils
    anchors = _gens()[0]((128, 128, 3), 4)
    for impl in (geometry.box_from_anchor_and_target, box_utils.box_from_anchor_and_target):
        out = impl(anchors, np.zeros_like(anchors))
        assert np.array_equal(out, anchors)
        big = np.zeros_like(anchors); big[:, 2] = 1000.0; big[:, 3] = -1000.0
        out = impl(anchors, big)
        assert np.allclose(out[:, 2], anchors[:, 2] * 1e4) and np.allclose(out[:, 3], anchors[:, 3] * 1e-4)


def test_iou_quirk_identical_box_is_not_one():
    """bbox_iou_vuvu uses (min-max+1) for the areas (box_utils.py:140-141): documented, not fixed."""
    from oracle import geometry
    from bayes_od_rc_amd import box_utils
    b = np.array([[10.0, 20.0, 50.0, 100.0]], np.float32)     # h = 40, w = 80
    h, w = 40.0, 80.0
    expect = (w + 1) * (h + 1) / (2 * (w - 1) * (h - 1) - (w + 1) * (h + 1) + 1e-5)
    for impl in (geometry.bbox_iou_vuvu, box_utils.bbox_iou_vuvu):
        got = impl(b, b)[0, 0]
        assert abs(got - expect) < 1e-5 and got > 1.0
    far = np.array([[500.0, 500.0, 520.0, 520.0]], np.float32)
    assert geometry.bbox_iou_vuvu(b, far)[0, 0] == 0.0


def test_targets_round_trip():
    from oracle import geometry
    from bayes_od_rc_amd.anchor_generator import FpnAnchorGenerator
    anchors = np.concatenate([g for g in [_gens()[0]((256, 256, 3), l) for l in (3, 4, 5)]])
    gt = np.array([[100.0, 120.0, 60.0, 40.0], [40.0, 200.0, 30.0, 90.0]], np.float32)   # vuhw
    gt_cls = np.eye(8, dtype=np.float32)[[0, 3]]
    ious = geometry.bbox_iou_vuvu(geometry.vuhw_to_vuvu(anchors), geometry.vuhw_to_vuvu(gt))
    for batching, targets in ((geometry.positive_negative_batching, geometry.generate_anchor_targets),
                              (FpnAnchorGenerator.positive_negative_batching, FpnAnchorGenerator.generate_anchor_targets)):
        pos, neg, arg = batching(ious, 0.5, 0.4)
        assert pos.sum() > 0 and not np.any(pos & neg)
        box_t, cls_t = targets(anchors, gt, gt_cls, arg, pos)
        rec = geometry.box_from_anchor_and_target(anchors[pos], box_t[pos].astype(np.float32))
        assert np.allclose(rec, gt[arg[pos]], rtol=1e-4, atol=1e-3)     # demo's reconstruction (anchor_generation_demo.py:104-105)
        assert np.all(cls_t[~pos, 7] == 1) and np.all(cls_t[~pos, :7] == 0)
