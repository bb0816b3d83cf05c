"""GPU: the training step (bod_train_step) against oracle/torch_train.py -- the same
network, losses, dropout masks and optimizer under torch.autograd in float64 on the CPU.

Two handles, one executor (csrc/train_impl.inc).  The fp32 training handle (precision='fp32': fp32 tensors, exact-fp32 MFMA
GEMMs) meets the oracle ELEMENT-WISE on every gradient tensor, the loss terms and the first optimizer update.  The bf16
handle (the product: bf16 activations / weights / dZ, fp32 accumulation through ~60 layers) runs the same op walk and is
compared per tensor by direction and size (cosine similarity, norm ratio): bf16 storage decorrelates two evaluations of
this network at the level of its own rounding noise (oracle/torch_train.py's bf16 emulation accumulated in fp32 vs in
fp64 already differ by ~10 % per tensor), so no element-wise bound exists for it."""
import numpy as np
import pytest

from conftest import ANCHOR_CFG

pytestmark = pytest.mark.gpu


def _problem(hw=(64, 64), batch=2, seed=0, depth=50):
    from bayes_od_rc_amd import synthetic
    from bayes_od_rc_amd.anchor_generator import FpnAnchorGenerator
    rng = np.random.default_rng(seed)
    weights = synthetic.make_weights(cls_fg_bias=-2.0, depth=depth)
    anchors = FpnAnchorGenerator(ANCHOR_CFG).generate_all((hw[0], hw[1], 3)).astype(np.float32)
    a = anchors.shape[0]
    frames = synthetic.make_frames(batch, hw[0], hw[1], seed=5)
    pos = (rng.uniform(size=(batch, a)) < 0.05).astype(np.uint8)
    neg = ((rng.uniform(size=(batch, a)) < 0.5) & (pos == 0)).astype(np.uint8)
    cls_t = np.zeros((batch, a, 8), np.float32)
    cls_t[..., 7] = 1.0
    fg = rng.integers(0, 7, size=(batch, a))
    for b in range(batch):
        idx = np.nonzero(pos[b])[0]
        cls_t[b, idx, 7] = 0.0
        cls_t[b, idx, fg[b, idx]] = 1.0
    box_t = rng.normal(0, 0.5, (batch, a, 4)).astype(np.float32)
    return weights, anchors, frames, cls_t, box_t, pos, neg


def _cos(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


@pytest.mark.parametrize("depth", [50, 101])
def test_gradients_match_autograd(depth):
    from bayes_od_rc_amd.engine import Engine, make_config
    from oracle import torch_train
    hw, batch = (64, 64), 2
    weights, anchors, frames, cls_t, box_t, pos, neg = _problem(hw, batch, depth=depth)
    eng = Engine(make_config(hw, batch=batch, mc_samples=1, training=True, backbone_depth=depth))
    eng.load_weights(weights)
    eng.set_anchors(anchors)
    got = eng.train_step(frames, cls_t, box_t, pos, neg, seed=3, first_image_id=10, apply_update=False)
    ref, grads, _, _ = torch_train.train_step(weights, frames, cls_t, box_t, anchors, pos, neg, seed=3, first_image_id=10)
    for k in ("total_loss", "cls_loss", "reg_loss", "covariance_loss", "regularization_loss"):
        assert abs(got[k] - ref[k]) <= 2e-2 * abs(ref[k]) + 1e-6, (k, got[k], ref[k])
    assert abs(got["grad_norm"] - ref["grad_norm"]) < 0.1 * ref["grad_norm"]
    worst = {}
    for name, g in grads.items():
        layer, kind = name.rsplit("/", 1)
        if layer == "pyramid_regression_3":
            continue                          # constructed but never called (a4): not part of the model's variables
        mine = eng.train_get(layer, kind, g.shape, what="grad")
        nr = np.linalg.norm(g)
        if nr < 1e-7 * ref["grad_norm"]:
            continue
        worst[name] = (_cos(mine, g), float(np.linalg.norm(mine) / nr))
    # bf16 storage through ~60 layers on 64x64 inputs (2x2 / 1x1 maps at the top of the pyramid): the worst tensors sit at
    # cosine 0.975; a wrong term anywhere in the backward pass shows as a cosine near 0 or a norm ratio far from 1
    bad = {k: v for k, v in worst.items() if v[0] < 0.95 or not (0.85 < v[1] < 1.15)}
    assert not bad, sorted(bad.items(), key=lambda kv: kv[1][0])[:8]
    assert len(worst) > (200 if depth == 50 else 330)
    cosines = np.array([v[0] for v in worst.values()])
    assert np.median(cosines) > 0.99, float(np.median(cosines))


@pytest.mark.parametrize("depth", [50, 101])
def test_fp32_training_handle_gradients_match_autograd_elementwise(depth):
    """Every gradient tensor of the whole step (backbone, FPN, towers, heads; kernels, biases, BatchNorm gamma / beta),
    element by element, against float64 autograd: |d| <= 1e-4 max|g| per tensor (measured 3e-6), loss terms to 1e-5."""
    from bayes_od_rc_amd.engine import Engine, make_config
    from oracle import torch_train
    hw, batch = (64, 64), 2
    weights, anchors, frames, cls_t, box_t, pos, neg = _problem(hw, batch, depth=depth)
    eng = Engine(make_config(hw, batch=batch, mc_samples=1, training=True, backbone_depth=depth, precision="fp32"))
    eng.load_weights(weights)
    eng.set_anchors(anchors)
    got = eng.train_step(frames, cls_t, box_t, pos, neg, seed=3, first_image_id=10, apply_update=False)
    ref, grads, _, _ = torch_train.train_step(weights, frames, cls_t, box_t, anchors, pos, neg, seed=3, first_image_id=10)
    for k in ("total_loss", "cls_loss", "reg_loss", "covariance_loss", "regularization_loss", "grad_norm"):
        assert abs(got[k] - ref[k]) <= 1e-5 * abs(ref[k]) + 1e-7, (k, got[k], ref[k])
    checked, worst = 0, (0.0, None)
    for name, g in grads.items():
        layer, kind = name.rsplit("/", 1)
        if layer == "pyramid_regression_3":
            continue                          # constructed but never called (a4): not part of the model's variables
        mine = eng.train_get(layer, kind, g.shape, what="grad").astype(np.float64)
        scale = np.abs(g).max()
        if scale < 1e-9 * ref["grad_norm"]:
            assert np.abs(mine).max() <= 1e-7 * ref["grad_norm"], name
            continue
        err = float(np.abs(mine - g).max() / scale)
        worst = max(worst, (err, name))
        assert err <= 1e-4, (name, err)
        checked += 1
    assert checked > (250 if depth == 50 else 400), checked
    print("worst element-wise gradient error / max|g|: %.2e (%s), %d tensors" % (worst[0], worst[1], checked))


def test_fp32_training_handle_first_update_matches_the_oracle_elementwise():
    """Global-norm clip + keras Adam(epsilon 1e-2) on the fp32 handle: first moments and updated parameters element-wise."""
    from bayes_od_rc_amd.engine import Engine, make_config
    from oracle import torch_train
    hw, batch = (64, 64), 2
    weights, anchors, frames, cls_t, box_t, pos, neg = _problem(hw, batch, seed=1)
    eng = Engine(make_config(hw, batch=batch, mc_samples=1, training=True, precision="fp32"))
    eng.load_weights(weights)
    eng.set_anchors(anchors)
    eng.train_step(frames, cls_t, box_t, pos, neg, seed=3, first_image_id=10, learning_rate=1e-3)
    _, _, new_w, state = torch_train.train_step(weights, frames, cls_t, box_t, anchors, pos, neg, seed=3, first_image_id=10, lr=1e-3)
    checked = 0
    for name, w_new in new_w.items():
        layer, kind = name.rsplit("/", 1)
        if layer == "pyramid_regression_3":
            continue
        m_ref = state[name][0].detach().numpy()
        m_ref = np.transpose(m_ref, (2, 3, 1, 0)) if kind == "kernel" else m_ref
        if np.abs(m_ref).max() < 1e-12:
            continue
        m_got = eng.train_get(layer, kind, w_new.shape, what="adam_m").astype(np.float64)
        assert np.abs(m_got - m_ref).max() <= 1e-4 * np.abs(m_ref).max(), name
        old = np.asarray(weights[layer][kind], np.float64)
        d_ref = w_new - old
        d_got = eng.train_get(layer, kind, w_new.shape).astype(np.float64) - old
        # the update against fp32 resolution of the parameter it is added to (1e-3 * lr-sized steps on O(0.1) weights)
        assert np.abs(d_got - d_ref).max() <= 1e-4 * np.abs(d_ref).max() + 2e-7 * max(np.abs(old).max(), 1e-3), (name, np.abs(d_got - d_ref).max(), np.abs(d_ref).max())
        checked += 1
    assert checked > 250, checked
    # and the handle keeps descending on a fixed batch
    losses = [eng.train_step(frames, cls_t, box_t, pos, neg, seed=3, first_image_id=10, learning_rate=1e-3)["total_loss"] for _ in range(8)]
    assert losses[-1] < losses[0], losses


@pytest.mark.parametrize("graph,threads,wgrad_streams", [("0", "1", "2"), ("1", "1", "2"), ("0", "2", "2"), ("0", "1", "1")])
def test_one_step_update_and_descent(graph, threads, wgrad_streams, monkeypatch):
    """Adam(epsilon 1e-2) with global-norm clipping: the first update matches the oracle's in direction and size for every
    tensor, and repeated steps on a fixed batch lower the loss."""
    from bayes_od_rc_amd.engine import Engine, make_config
    from oracle import torch_train
    monkeypatch.setenv("BOD_TRAIN_GRAPH", graph)      # 1: the step is recorded into a hipGraph on its second run and replayed
    monkeypatch.setenv("BOD_TRAIN_THREADS", threads)  # 2: the weight-gradient streams get their own enqueue thread
    monkeypatch.setenv("BOD_TRAIN_WGRAD_STREAMS", wgrad_streams)   # weight-gradient streams (2 = one per half of the dZ^T double buffer)
    hw, batch = (64, 64), 2
    weights, anchors, frames, cls_t, box_t, pos, neg = _problem(hw, batch, seed=1)
    eng = Engine(make_config(hw, batch=batch, mc_samples=1, training=True))
    eng.load_weights(weights)
    eng.set_anchors(anchors)
    first = eng.train_step(frames, cls_t, box_t, pos, neg, seed=3, first_image_id=10, learning_rate=1e-3)
    _, _, new_w, state = torch_train.train_step(weights, frames, cls_t, box_t, anchors, pos, neg, seed=3, first_image_id=10, lr=1e-3)
    checked = big = 0
    for name, w_new in new_w.items():
        layer, kind = name.rsplit("/", 1)
        if layer == "pyramid_regression_3":
            continue
        m_ref = state[name][0].detach().numpy()
        m_ref = np.transpose(m_ref, (2, 3, 1, 0)) if kind == "kernel" else m_ref
        if np.linalg.norm(m_ref) < 1e-12:
            continue
        # first moment after one step = (1 - beta1) * clipped gradient: checks the clip factor and the Adam bookkeeping
        m_got = eng.train_get(layer, kind, w_new.shape, what="adam_m")
        assert _cos(m_got, m_ref) > 0.95 and 0.85 < np.linalg.norm(m_got) / np.linalg.norm(m_ref) < 1.15, name
        checked += 1
        # the weight update itself where it is well above fp32 resolution of the weights
        old = np.asarray(weights[layer][kind], np.float64)
        d_ref = w_new - old
        sel = np.abs(d_ref) > 2e-5 * np.maximum(np.abs(old), 1e-3)
        if sel.sum() >= 16:
            d_got = eng.train_get(layer, kind, w_new.shape).astype(np.float64) - old
            assert _cos(d_got[sel], d_ref[sel]) > 0.9, (name, _cos(d_got[sel], d_ref[sel]))
            big += 1
    assert checked > 200 and big > 5, (checked, big)
    losses = [first["total_loss"]]
    for i in range(12):
        losses.append(eng.train_step(frames, cls_t, box_t, pos, neg, seed=3, first_image_id=10, learning_rate=1e-3)["total_loss"])
    assert losses[-1] < 0.8 * losses[0], losses
    # the forward pass of the inference API runs on the updated weights (same handle)
    eng.forward(frames, seed=3, first_image_id=10)
    assert np.isfinite(eng.get_raw()[0]).all()


def test_run_training_cli(tmp_path, monkeypatch):
    """run_training with the reference's flags on synthetic samples: the yaml's losses / minibatch / schedule drive
    bod_train_step, the loss goes down, and the .npz checkpoint loads into the inference model."""
    import os
    monkeypatch.setenv("BAYESOD_DATA_DIR", str(tmp_path))
    from bayes_od_rc_amd import run_training
    from bayes_od_rc_amd.model import RetinaNetModel
    from bayes_od_rc_amd import config_utils
    history, ckpt_dir = run_training.main(["--gpu_device", "0", "--synthetic", "3", "--image_size", "128", "128", "--steps", "10"])
    assert len(history) == 10 and np.isfinite(history).all() and history[-1] < history[0]
    path = os.path.join(ckpt_dir, "ckpt-10.npz")
    assert os.path.exists(path)
    here = os.path.dirname(os.path.abspath(run_training.__file__))
    cfg = config_utils.load_yaml(os.path.join(here, "configs", "retinanet_bdd_covar.yaml"))

    class A(object):
        data_split = "test"
        yaml_path = os.path.join(here, "configs", "retinanet_bdd_covar.yaml")
    cfg = config_utils.setup(cfg, A())
    model = RetinaNetModel(cfg["model_config"])
    model.load_weights(path)
    from bayes_od_rc_amd import synthetic
    pred = model(synthetic.make_frames(1, 128, 128, seed=1), train_val_test="validation")
    assert np.isfinite(pred["anchors_class_predictions"]).all()


def test_run_training_resumes_from_the_latest_checkpoint(tmp_path, monkeypatch):
    """ckpt.restore(manager.latest_checkpoint) (run_training.py:75-82): a second invocation continues at the saved step with
    the saved Adam moments, update counter, learning-rate position and dropout image ids; the resumed run tracks an
    uninterrupted one (bf16 storage + atomic gradient adds: to a few per cent, not bit for bit)."""
    import os
    from bayes_od_rc_amd import run_training
    common = ["--gpu_device", "0", "--synthetic", "3", "--image_size", "128", "128"]
    monkeypatch.setenv("BAYESOD_DATA_DIR", str(tmp_path / "whole"))
    whole, _ = run_training.main(common + ["--steps", "8"])
    monkeypatch.setenv("BAYESOD_DATA_DIR", str(tmp_path / "split"))
    first, ckpt_dir = run_training.main(common + ["--steps", "4"])
    z = np.load(os.path.join(ckpt_dir, "ckpt-4.npz"))
    assert int(z["__optimizer__/optimizer/step"]) == 4 and int(z["__optimizer__/trainer/step"]) == 4
    assert "__optimizer__/pyramid_classification_0/kernel/adam_m" in z.files and np.abs(z["__optimizer__/P3/kernel/adam_v"]).max() > 0
    second, _ = run_training.main(common + ["--steps", "8"])
    assert len(first) == 4 and len(second) == 4                 # resumed at step 4, not at 0
    assert os.path.exists(os.path.join(ckpt_dir, "ckpt-8.npz"))
    assert int(np.load(os.path.join(ckpt_dir, "ckpt-8.npz"))["__optimizer__/optimizer/step"]) == 8
    assert np.allclose(first, whole[:4], rtol=2e-2)
    assert np.allclose(second, whole[4:], rtol=8e-2), (second, whole[4:])
    # a restart from scratch would begin with the initial loss again
    assert second[0] < 0.9 * whole[0]
    before = {f: os.path.getsize(os.path.join(ckpt_dir, f)) for f in os.listdir(ckpt_dir) if f.startswith("ckpt-")}
    fresh, _ = run_training.main(common + ["--steps", "2", "--no_resume"])
    assert abs(fresh[0] - whole[0]) < 2e-2 * abs(whole[0])
    # --no_resume never deletes: the previous run's checkpoints were moved aside, whole, and the fresh run's own is the latest
    aside = [d for d in os.listdir(ckpt_dir) if d.startswith("superseded-")]
    assert len(aside) == 1
    kept = {f: os.path.getsize(os.path.join(ckpt_dir, aside[0], f)) for f in os.listdir(os.path.join(ckpt_dir, aside[0]))}
    assert kept == before and len(before) >= 2
    assert run_training.latest_checkpoint(ckpt_dir).endswith("ckpt-2.npz")
    # ... and not before the trainer exists: a run that fails on its weight file leaves the directory as it was
    with pytest.raises(Exception):
        run_training.main(common + ["--steps", "2", "--no_resume", "--weights", str(tmp_path / "missing.npz")])
    assert run_training.latest_checkpoint(ckpt_dir).endswith("ckpt-2.npz") and len([d for d in os.listdir(ckpt_dir) if d.startswith("superseded-")]) == 1


def test_learning_rate_schedule():
    from bayes_od_rc_amd.run_training import piecewise_learning_rate
    lr = piecewise_learning_rate({"initial_learning_rate": 0.001, "decay_boundaries": [3, 9], "decay_factor": 0.1}, epoch_size=100)
    assert lr(0) == 0.001 and lr(300) == 0.001 and lr(301) == 0.0001 and lr(900) == 0.0001 and abs(lr(901) - 1e-5) < 1e-12


def test_train_apply_equals_fused_update():
    """bod_train_step(apply_update=0) + bod_train_apply == bod_train_step(apply_update=1) (world size 1 of the data-parallel
    path)."""
    from bayes_od_rc_amd.engine import Engine, make_config
    from bayes_od_rc_amd import distributed as bd
    hw, batch = (64, 64), 2
    weights, anchors, frames, cls_t, box_t, pos, neg = _problem(hw, batch, seed=2)
    engs = []
    for _ in range(2):
        e = Engine(make_config(hw, batch=batch, mc_samples=1, training=True))
        e.load_weights(weights)
        e.set_anchors(anchors)
        engs.append(e)
    a = engs[0].train_step(frames, cls_t, box_t, pos, neg, seed=3, first_image_id=10, learning_rate=1e-3)
    b = bd.data_parallel_train_step(engs[1], frames, cls_t, box_t, pos, neg, 1e-3, seed=3, first_image_id=10)
    # two runs of the same step differ in the last bits (fp32 atomics in the residual scatter / col2im / dot products)
    assert abs(a["total_loss"] - b["total_loss"]) < 1e-6 * a["total_loss"] and abs(a["grad_norm"] - b["grad_norm"]) < 1e-4 * a["grad_norm"]
    for layer, kind, shape in (("pyramid_cov_2", "kernel", (3, 3, 256, 256)), ("res2a_branch1", "bias", (256,)), ("bn5c_branch2c", "beta", (2048,))):
        m0, m1 = engs[0].train_get(layer, kind, shape, what="adam_m"), engs[1].train_get(layer, kind, shape, what="adam_m")
        # (an fp32 atomic-order difference early in the backward pass can flip a bf16 rounding of dZ further down)
        assert np.abs(m0 - m1).max() <= 1e-2 * np.abs(m0).max() + 1e-12, layer
        w0, w1 = engs[0].train_get(layer, kind, shape), engs[1].train_get(layer, kind, shape)
        assert np.abs(w0 - w1).max() <= 1e-6, layer


def test_data_parallel_two_ranks_on_one_gpu():
    """Two ranks on this box's GPU (gloo): one all-reduce of the gradient arena per step; step 1 equals the hand-averaged
    single-process update bit for bit, ranks stay identical afterwards."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(root, "tests", "tools", "dp_train_worker.py")],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "DP_TRAIN_OK" in out.stdout


@pytest.mark.parametrize("hw,batch,depth", [((512, 512), 3, 50), ((512, 512), 3, 101)])
def test_full_size_step_descends(hw, batch, depth):
    """The yaml's minibatch at the BASELINE geometry (config 5: depth 101), where autograd on the host is too slow:
    the losses and the gradient norm stay finite, repeated steps on one batch lower the loss, and the inference API
    of the same handle runs on the updated weights."""
    from bayes_od_rc_amd.engine import Engine, make_config
    weights, anchors, frames, cls_t, box_t, pos, neg = _problem(hw, batch, seed=2, depth=depth)
    eng = Engine(make_config(hw, batch=batch, mc_samples=1, training=True, backbone_depth=depth))
    eng.load_weights(weights)
    eng.set_anchors(anchors)
    eng.upload_images(frames)
    out = [eng.train_step(None, cls_t, box_t, pos, neg, seed=4, first_image_id=0, learning_rate=1e-3) for _ in range(10)]
    for o in out:
        assert all(np.isfinite(v) for v in o.values()), o
    assert out[-1]["total_loss"] < 0.9 * out[0]["total_loss"], [o["total_loss"] for o in out]
    eng.forward(frames, seed=4, first_image_id=0)
    assert np.isfinite(eng.get_raw()[0]).all()
    eng.close()


def test_run_validation_over_training_checkpoints(tmp_path, monkeypatch):
    """run_training writes checkpoints, run_validation evaluates each of them once (run_validation.py:86-228): losses,
    post-processed predictions in the BDD json layout, evaluated_ckpts.txt bookkeeping, nothing to do on a second pass."""
    import json
    import os
    monkeypatch.setenv("BAYESOD_DATA_DIR", str(tmp_path))
    from bayes_od_rc_amd import run_training, run_validation
    history, ckpt_dir = run_training.main(["--gpu_device", "0", "--synthetic", "3", "--image_size", "128", "128", "--steps", "6"])
    assert os.path.exists(os.path.join(ckpt_dir, "ckpt-6.npz"))
    res = run_validation.main(["--gpu_device", "0", "--synthetic", "2", "--image_size", "128", "128"])
    assert [r["ckpt_id"] for r in res] == sorted(r["ckpt_id"] for r in res) and res[-1]["ckpt_id"] == 6
    for r in res:
        assert r["num_frames"] == 2 and np.isfinite(r["mean_total_loss"]) and set(r["mean_losses"]) >= {"cls_loss", "reg_loss"}
    pred_root = os.path.join(os.path.dirname(ckpt_dir), "predictions")
    with open(os.path.join(pred_root, "validation", "6", "data", "predictions.json")) as fp:
        records = json.load(fp)
    assert len(records) == res[-1]["num_detections"]
    for rec in records[:5]:
        assert set(rec) >= {"name", "category", "bbox", "score"} and len(rec["bbox"]) == 4
    assert list(run_validation.get_evaluated_ckpts(pred_root)) == [r["ckpt_id"] for r in res]
    assert run_validation.main(["--gpu_device", "0", "--synthetic", "2", "--image_size", "128", "128"]) == []
