"""BASELINE config 1: anchor_generation_demo plumbing on CPU with a synthetic 512x512 frame."""
import os

import numpy as np

from conftest import ANCHOR_CFG, ROOT


def test_anchor_generation_demo(tmp_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("anchor_demo", os.path.join(ROOT, "demos", "anchor_generation_demo.py"))
    demo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo)
    out = tmp_path / "a.png"
    res = demo.main(["--out", str(out)])
    assert out.exists() and out.stat().st_size > 1000
    s = res["sample"]
    assert s["anchors"].shape == (49104, 4) and s["image_normalized"].shape == (512, 512, 3)
    assert s["anchors_box_targets"].shape == (49104, 4) and s["anchors_class_targets"].shape == (49104, 8)
    pos, neg = s["positive_anchors_mask"], s["negative_anchors_mask"]
    assert pos.sum() > 0 and not np.any(pos & neg)
    # every reconstructed box equals one of the synthetic GT boxes (anchor_generation_demo.py:104-122)
    rec = res["reconstructed_gt_corners"]
    d = np.abs(rec[:, None, :] - res["gt_vuvu"][None]).max(axis=2).min(axis=1)
    assert d.max() < 1e-2
    # class targets of positives are the GT one-hots, never background
    assert np.all(res["positive_classes"][:, 7] == 0) and np.all(res["positive_classes"].sum(1) == 1)
    # background one-hot everywhere else
    assert np.all(s["anchors_class_targets"][~pos][:, 7] == 1)


def test_sample_dict_testing_mode_has_anchors_only():
    from bayes_od_rc_amd import sample_builder, synthetic
    s = sample_builder.create_sample_dict(synthetic.make_frames(1, 128, 160)[0], ANCHOR_CFG, is_testing=True)
    assert set(s) == {"image_normalized", "im_size", "anchors"}
    assert s["anchors"].shape[0] == 9 * (16 * 20 + 8 * 10 + 4 * 5 + 2 * 3 + 1 * 2)
