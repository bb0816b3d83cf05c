"""Host-side mirror of the reference surface that needs no GPU: config plumbing, error behaviour,
model construction, writers' directory layout."""
import argparse
import os

import numpy as np
import pytest

from conftest import ROOT


def _cfg():
    from bayes_od_rc_amd import config_utils
    return config_utils.load_yaml(os.path.join(ROOT, "bayes-od-rc_amd", "configs", "retinanet_bdd_covar.yaml"))


def test_setup_injects_derived_fields(tmp_path, monkeypatch):
    from bayes_od_rc_amd import config_utils
    monkeypatch.setenv("BAYESOD_DATA_DIR", str(tmp_path))
    path = os.path.join(ROOT, "bayes-od-rc_amd", "configs", "retinanet_bdd_covar.yaml")
    cfg = config_utils.setup(_cfg(), argparse.Namespace(yaml_path=path, data_split="test"))
    assert cfg["model_config"]["header"]["num_classes"] == 7
    assert cfg["model_config"]["header"]["anchors_per_location"] == 9
    assert cfg["dataset_config"]["data_split"] == "test"
    assert os.path.isdir(tmp_path / "outputs" / "retinanet_bdd_covar" / "checkpoints")
    assert os.path.isfile(tmp_path / "outputs" / "retinanet_bdd_covar" / "retinanet_bdd_covar.yaml")
    bad = tmp_path / "other_name.yaml"
    bad.write_text(open(path).read())
    with pytest.raises(ValueError):
        config_utils.setup(_cfg(), argparse.Namespace(yaml_path=str(bad), data_split="test"))


def test_model_constructor_errors():
    from bayes_od_rc_amd.model import RetinaNetModel, fill_triangular_4
    mc = _cfg()["model_config"]
    with pytest.raises(ValueError):
        RetinaNetModel(mc)                                   # setup() not run: no num_classes
    mc["header"].update(num_classes=7, anchors_per_location=9)
    m = RetinaNetModel(mc)
    assert m.mc_dropout_samples == 10 and m.compute_covar
    with pytest.raises(ValueError):
        m(np.zeros((1, 128, 128, 3), np.float32))            # no weights loaded
    with pytest.raises(ValueError):
        m(np.zeros((128, 128, 3), np.float32))
    mc2 = dict(mc, output_names=["classification"])
    with pytest.raises(ValueError):
        RetinaNetModel(mc2)
    x = np.arange(10.0)
    assert fill_triangular_4(x).tolist() == [[4, 0, 0, 0], [8, 9, 0, 0], [7, 6, 5, 0], [3, 2, 1, 0]]


def test_prediction_writer_layout(tmp_path):
    from bayes_od_rc_amd import writers
    w = writers.PredictionWriter(str(tmp_path), "bdd", 101, "bayes_od", "none")
    assert w.root.endswith(os.path.join("testing", "bdd", "101", "bayes_od_none"))
    boxes = np.array([[1.0, 2.0, 30.0, 40.0]], np.float32)
    cls = np.array([[0.7, 0.1, 0.05, 0.05, 0.04, 0.03, 0.02, 0.01]], np.float32)
    w.write("000000", boxes, cls, boxes, np.eye(4)[None], cls, cls * 31,
            ['car', 'truck', 'bus', 'person', 'rider', 'bike', 'motor'])
    w.close()
    for sub in ("mean", "cov", "cat_param", "cat_count"):
        assert os.path.isfile(os.path.join(w.root, sub, "000000.npy"))
    import json
    rec = json.load(open(os.path.join(w.root, "data", "predictions.json")))
    assert rec[0]["category"] == "car" and rec[0]["bbox"] == [2.0, 1.0, 40.0, 30.0] and rec[0]["timestep"] == 1000


def test_synthetic_inputs_follow_reference_convention():
    from bayes_od_rc_amd import synthetic
    f = synthetic.make_frames(2, 32, 48, seed=5)
    assert f.shape == (2, 32, 48, 3) and f.dtype == np.float32
    rgb = np.random.default_rng(5).integers(0, 256, size=(32, 48, 3), dtype=np.uint8).astype(np.float32)
    assert np.array_equal(f[0][..., 0], rgb[..., 2] - np.float32(103.94))      # BGR, ImageNet means
    assert np.array_equal(f[0][..., 2], rgb[..., 0] - np.float32(123.68))
    w = synthetic.make_weights()
    assert w["conv1"]["kernel"].shape == (7, 7, 3, 64)
    assert w["res5c_branch2c"]["kernel"].shape == (1, 1, 512, 2048)
    assert w["pyramid_classification"]["kernel"].shape == (1, 1, 256, 72)
    b = w["pyramid_classification"]["bias"].reshape(9, 8)
    assert np.allclose(b[:, :7], -np.log(99.0)) and np.all(b[:, 7] == 0)        # multitask_headers.py:79-83
    assert "pyramid_regression_3" in w                                          # built but unused


def test_checkpoint_files_are_atomic_and_resume_skips_damaged_ones(tmp_path):
    """run_training's checkpoint helpers (reference: tf.train.CheckpointManager, run_training.py:72-82): a save goes through a
    temp file + rename, names that are not ckpt-<int>.npz are ignored, and listing orders by step."""
    from bayes_od_rc_amd import run_training as rt

    class FakeTrainer:
        def weights(self):
            return {"conv1": {"kernel": np.ones((2, 2), np.float32), "bias": None}}

        def optimizer_state(self):
            return {"optimizer/step": np.int64(3), "trainer/step": np.int64(3)}

    d = str(tmp_path)
    rt.save_checkpoint(FakeTrainer(), os.path.join(d, "ckpt-3.npz"))
    rt.save_checkpoint(FakeTrainer(), os.path.join(d, "ckpt-12.npz"))
    open(os.path.join(d, "ckpt-foo.npz"), "w").close()                 # stray name: must not raise
    with open(os.path.join(d, "ckpt-20.npz"), "wb") as fp:              # truncated "latest": unreadable
        fp.write(b"PK\x03\x04 not a zip")
    assert [os.path.basename(p) for p in rt.sorted_checkpoints(d)] == ["ckpt-3.npz", "ckpt-12.npz", "ckpt-20.npz"]
    assert not [f for f in os.listdir(d) if f.startswith(".tmp-")]
    with np.load(os.path.join(d, "ckpt-12.npz")) as z:
        assert set(z.files) == {"conv1/kernel", rt.OPT_PREFIX + "optimizer/step", rt.OPT_PREFIX + "trainer/step"}
    assert rt.sorted_checkpoints(os.path.join(d, "missing")) == []
