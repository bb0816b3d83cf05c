"""GPU: the PRODUCTION instantiations of the hot kernel against the oracle, directly.

The tile rule picks `conv_igemm_kernel<256,256,2,4,0,true>` (activation row reuse + fused 1x1 head outputs) and the
256x256 N-way fan-out epilogue only once a launch has enough tiles, which none of the oracle-sized inputs reach.
`BOD_FORCE_CONV_TILE=256` (read once per process) plans exactly those instantiations on small inputs, so the tests
below run the oracle comparisons in a child pytest process with that switch set:

  * heads on the device's own pyramid vs the oracle's bf16-storage emulation (4 layers deep: tight bound)
  * whole forward at the bf16 noise floor, the stage chain (posterior / soft-NMS / clustering) on the device's outputs
  * one head-shaped conv and the dropout contract through bod_stage_conv (the <...,false> 256 tile)

and the bf16 pipeline's distance to the float64 pipeline at DETECTION level (final mu / Sigma / scores).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ANCHOR_CFG, BAYES_CFG, NMS_CFG, ROOT

pytestmark = pytest.mark.gpu


def _rms(x):
    return float(np.sqrt((np.asarray(x, np.float64) ** 2).mean()))


def test_heads_on_device_pyramid_match_bf16_emulation():
    """Head towers (fan-out layer 0, per-sample layers 1..3, 1x1 outputs) on the DEVICE's pyramid vs the oracle's
    bf16-storage emulation fed the same pyramid and the same Philox masks: only fp32 summation order and the rare 1-ulp
    bf16 flips it causes separate the two, four layers deep."""
    from bayes_od_rc_amd import synthetic
    from bayes_od_rc_amd.engine import Engine, make_config
    from oracle import network, philox
    hw, batch, n, seed, first = (128, 160), 2, 3, 424242, 9
    w = synthetic.make_weights()
    frames = synthetic.make_frames(batch, hw[0], hw[1], seed=12)
    eng = Engine(make_config(hw, batch=batch, mc_samples=n))
    eng.load_weights(w)
    eng.forward(frames, seed=seed, first_image_id=first)
    cls, box, cov = eng.get_raw()
    pyr = [eng.get_pyramid(l) for l in range(5)]
    nm = network.make_numerics(w, "bf16")
    for b in range(batch):
        km = lambda s, lid: philox.dropout_keep_mask(seed, first + b, s, lid, eng.P, 256, 0.3)
        level_maps = [p[b][None] for p in pyr]
        for got, head, c in ((cls[b], "cls", 8), (box[b], "reg", 4), (cov[b], "cov", 10)):
            ref = network.head_tower(nm, level_maps, head, n, km, 0.3, c)
            assert got.shape == ref.shape
            d = _rms(got - ref) / _rms(ref)
            worst = float(np.max(np.abs(got - ref))) / _rms(ref)
            print("heads-on-device-pyramid %s: rel RMS %.2e, max/RMS %.2e" % (head, d, worst))
            assert d < 4e-3, (head, d)
            assert worst < 6e-2, (head, worst)
    eng.close()


FORCED_SUITE = [
    "tests/test_gpu_pipeline.py::test_fp32_pipeline_matches_oracle_end_to_end[bf16x3]",      # bf16x3 on the fused row-reuse kernel + MC aggregation
    "tests/test_gpu_forward.py::test_fp32_mode_end_to_end[hw0-2-3-50-bf16x3]",
    "tests/test_gpu_production_kernel.py::test_heads_on_device_pyramid_match_bf16_emulation",
    "tests/test_gpu_forward.py::test_forward_at_bf16_noise_floor",
    "tests/test_gpu_pipeline.py::test_infer_stage_chain_matches_oracle",
    "tests/test_gpu_conv.py::test_conv_matches_oracle",
    "tests/test_gpu_conv.py::test_conv_dropout_matches_philox_contract",
    "tests/test_gpu_conv.py::test_conv_residual_relu_bf16_store",
]


def test_production_instantiations_meet_the_oracle_with_forced_256_tiles():
    """conv_igemm_kernel<256,256,2,4,0,true> + fused 1x1 + 256-tile fan-out against the oracle directly (not through a
    chain of self-comparisons): the oracle tests above and in the other files, re-run with BOD_FORCE_CONV_TILE=256."""
    env = dict(os.environ, BOD_FORCE_CONV_TILE="256")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + FORCED_SUITE,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=3000)
    tail = r.stdout[-3000:] + r.stderr[-2000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout and "skipped" not in r.stdout, tail


def test_forced_tile_really_plans_the_production_kernel():
    """Guard for the test above: with BOD_FORCE_CONV_TILE=256 the plan of a small handle uses the row-reuse tower kernel
    with fused outputs (the profiling hook `which=1` times only that kernel: it must see 4 launches per forward -- layer 1, layer 2
    of the continuing heads, layer 2 of the ending regression head, layer 3)."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from bayes_od_rc_amd import synthetic\n"
            "from bayes_od_rc_amd.engine import Engine, make_config\n"
            "eng = Engine(make_config((128, 128), batch=2, mc_samples=3))\n"
            "eng.load_weights(synthetic.make_weights())\n"
            "eng.upload_images(synthetic.make_frames(2, 128, 128, seed=1))\n"
            "eng.profile_begin(which=1)\n"
            "eng.forward(None, seed=1, first_image_id=0)\n"
            "p = eng.profile_end()\n"
            "print('XR_LAUNCHES', p['head_conv_launches'])\n" % ROOT)
    for forced, expect in (("256", "XR_LAUNCHES 4"), ("128", "XR_LAUNCHES 0")):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BOD_FORCE_CONV_TILE=forced),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert expect in r.stdout, (forced, r.stdout[-500:])


_AGG_SCRIPT = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from conftest import ANCHOR_CFG, BAYES_CFG, NMS_CFG
from bayes_od_rc_amd import synthetic
from bayes_od_rc_amd.anchor_generator import FpnAnchorGenerator
from bayes_od_rc_amd.engine import Engine, make_config
h, w, batch, n = (int(v) for v in sys.argv[3:7])
precision = sys.argv[7] if len(sys.argv) > 7 else "bf16"
eng = Engine(make_config((h, w), batch=batch, mc_samples=n, bayes_od_config=BAYES_CFG, nms_config=NMS_CFG, use_full_covar=True, precision=precision))
eng.load_weights(synthetic.make_weights(cls_fg_bias=-1.0))
eng.set_anchors(FpnAnchorGenerator(ANCHOR_CFG).generate_all((h, w, 3)))
frames = synthetic.make_frames(batch, h, w, seed=31)
eng.infer(frames, seed=77, first_image_id=5)
out = {"agg_plan": np.int32(eng.aggregating)}
for b in range(batch):
    post = eng.get_posterior(b)
    for k, v in post.items():
        out["post%d_%s" % (b, k)] = v
    for k, v in zip(("scores", "means", "covs", "counts"), eng.get_detections(b)):
        out["det%d_%s" % (b, k)] = v
raw_after_infer = [x.copy() for x in eng.get_raw()]          # after an aggregating infer: re-materialised
eng.forward(frames, seed=77, first_image_id=5)                 # the raw flavour itself
for name, a, b in zip(("cls", "box", "cov"), raw_after_infer, eng.get_raw()):
    assert np.array_equal(a, b), "raw %s differs between materialise-after-infer and forward" % name
    out["raw_" + name] = a
np.savez(sys.argv[2], **out)
"""


@pytest.mark.parametrize("h,w,batch,n,precision", [(128, 160, 2, 5, "bf16"), (96, 160, 1, 30, "bf16"), (128, 128, 3, 2, "bf16"), (160, 160, 1, 10, "bf16"),
                                                   (128, 160, 2, 5, "bf16x3"), (160, 160, 1, 10, "bf16x3"), (96, 96, 1, 30, "bf16x3"),
                                                   (128, 160, 2, 5, "f16mx"), (96, 96, 1, 30, "f16mx"), (128, 160, 2, 5, "f16mx4")])
def test_fused_mc_aggregation_equals_the_raw_path(tmp_path, h, w, batch, n, precision):
    """The MC aggregation fused into the last tower layers' epilogues (sum of softmax, Welford box mean / co-moments, sum of
    covariance parameters; sample-complete tiles) against the same pipeline with BOD_FUSE_AGGREGATION=0, i.e. raw
    [B,N,A,.] tensors + the posterior kernels' own loops: identical kept set, Dirichlet counts and scores bit for bit
    (the softmax sums are the same operations in the same order), box means / covariances to fp32 round-off (Welford
    vs two-pass), identical soft-NMS centres; and the raw tensors re-materialised after an aggregating infer equal a
    raw-flavour forward bit for bit.  Both precisions: the bf16x3 mode runs the same fused epilogue on (hi, lo) pairs
    (the production data path of the parity mode: no [B,N,A,.] tensors)."""
    outs = []
    for fuse in ("1", "0"):
        path = str(tmp_path / ("agg%s.npz" % fuse))
        env = dict(os.environ, BOD_FORCE_CONV_TILE="256", BOD_FUSE_AGGREGATION=fuse)
        r = subprocess.run([sys.executable, "-c", _AGG_SCRIPT, ROOT, path, str(h), str(w), str(batch), str(n), precision], env=env,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        z = np.load(path)
        outs.append({k: z[k] for k in z.files})
    fused, plain = outs
    assert int(fused["agg_plan"]) == 1 and int(plain["agg_plan"]) == 0       # the switch really selects the two plans (both precisions)
    for k in ("raw_cls", "raw_box", "raw_cov"):
        assert np.array_equal(fused[k], plain[k]), k               # the tiling of the last layers does not enter the arithmetic
    for b in range(batch):
        pre = "post%d_" % b
        assert fused[pre + "means"].shape[0] > 30
        assert np.array_equal(fused[pre + "anchor_index"], plain[pre + "anchor_index"])
        assert np.array_equal(fused[pre + "counts"], plain[pre + "counts"])
        assert np.array_equal(fused[pre + "score"], plain[pre + "score"])
        assert np.max(np.abs(fused[pre + "means"] - plain[pre + "means"]) / (np.abs(plain[pre + "means"]) + 1.0)) < 1e-4
        ref = plain[pre + "covs"]
        floor = np.abs(ref).reshape(len(ref), -1).max(axis=1)[:, None, None] * 1e-2
        assert (np.abs(fused[pre + "covs"] - ref) / (np.abs(ref) + floor)).max() < 5e-4          # Welford vs two-pass in fp32
        pre = "det%d_" % b
        assert fused[pre + "means"].shape == plain[pre + "means"].shape
        assert np.max(np.abs(fused[pre + "means"] - plain[pre + "means"]) / (np.abs(plain[pre + "means"]) + 1.0)) < 1e-4
        assert np.allclose(fused[pre + "scores"], plain[pre + "scores"], rtol=1e-5, atol=1e-6)


def _match(det_means, ref_means):
    """Greedy one-to-one matching of detections by IoU of their mean boxes (v,u,h,w) -> list of (i_det, i_ref, iou)."""
    from oracle import geometry
    if len(det_means) == 0 or len(ref_means) == 0:
        return []
    a, b = geometry.vuhw_to_vuvu(det_means.astype(np.float64)), geometry.vuhw_to_vuvu(ref_means.astype(np.float64))
    y1 = np.maximum(a[:, None, 0], b[None, :, 0]); x1 = np.maximum(a[:, None, 1], b[None, :, 1])
    y2 = np.minimum(a[:, None, 2], b[None, :, 2]); x2 = np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(y2 - y1, 0, None) * np.clip(x2 - x1, 0, None)
    area = lambda q: (q[:, 2] - q[:, 0]) * (q[:, 3] - q[:, 1])
    iou = inter / (area(a)[:, None] + area(b)[None, :] - inter + 1e-12)
    pairs = []
    while True:
        i, j = np.unravel_index(np.argmax(iou), iou.shape)
        if iou[i, j] < 0.5:
            break
        pairs.append((int(i), int(j), float(iou[i, j])))
        iou[i, :] = -1
        iou[:, j] = -1
    return pairs


@pytest.mark.parametrize("precision", ["bf16", "bf16x3", "f16mx", "f16mx4"])
def test_detection_level_distance_of_the_bf16_pipeline_to_float64(precision):
    """What bf16 storage costs at the OUTPUT of the path: final detections (cluster-fused mu, Sigma x70, class scores)
    of the benchmarked bf16 pipeline against the float64 oracle pipeline run from the same raw frames with the same
    Philox streams.  The candidate sets differ slightly (a 0.2-0.9 % perturbation of the head outputs moves a few
    categorical draws across a CDF boundary), so detections are matched by IoU and the bound is stated on the matches;
    the observed figures are printed and recorded in DESIGN.md section 6."""
    from bayes_od_rc_amd import synthetic
    from bayes_od_rc_amd.model import RetinaNetModel
    from bayes_od_rc_amd.anchor_generator import FpnAnchorGenerator
    from bayes_od_rc_amd.inference_utils import BayesOdPipeline
    from oracle import bayes_od, philox, network, nms, clustering, geometry
    hw, batch, n, seed = (160, 160), 2, 6, 2026
    cfg = {"output_names": ["classification", "regression", "regression_covar"], "mc_dropout_samples": n,
           "header": {"dropout_rate": 0.3, "num_classes": 7, "anchors_per_location": 9}}
    w = synthetic.make_weights(cls_fg_bias=-1.0)
    model = RetinaNetModel(cfg, precision=precision)
    model.load_weights(w)
    anchors = FpnAnchorGenerator(ANCHOR_CFG).generate_all((hw[0], hw[1], 3))
    pipe = BayesOdPipeline(model, hw, batch, BAYES_CFG, NMS_CFG, use_full_covar=True, anchors=anchors)
    frames = synthetic.make_frames(batch, hw[0], hw[1], seed=77)
    dets = pipe(frames, seed=seed, first_image_id=0)
    eng = pipe.engine
    stats = {"matched": 0, "dev": 0, "ref": 0, "dmu": [], "dcov": [], "dscore": [], "kept_jaccard": []}
    for b in range(batch):
        km = lambda s, lid: philox.dropout_keep_mask(seed, b, s, lid, eng.P, 256, 0.3)
        pred = network.retinanet_forward(w, frames[b][None], n, 8, mode="literal", dtype=np.float64, keep_masks=km)
        u = philox.categorical_uniforms(seed, b, eng.A)
        post = bayes_od.bayes_od_posterior(pred, anchors, u, BAYES_CFG, use_full_covar=True, dtype=np.float64)
        idx, _ = nms.soft_nms(post["corners"].astype(np.float32), post["ranking"].astype(np.float32), 100, 0.5, 0.5)
        iou = geometry.bbox_iou_vuvu(post["corners"], post["corners"])
        s, mu, cv, cn = clustering.bayes_od_clustering(post["counts"], post["means"], post["covs"], idx, iou, 0.5)
        scores, means, covs, counts = dets[b]
        got_keep = np.zeros(eng.A, bool)
        got_keep[eng.get_posterior(b)["anchor_index"]] = True
        stats["kept_jaccard"].append((got_keep & post["keep"]).sum() / max(1, (got_keep | post["keep"]).sum()))
        pairs = _match(means, mu[:, :, 0])
        stats["matched"] += len(pairs); stats["dev"] += len(means); stats["ref"] += len(mu)
        for i, j, _ in pairs:
            stats["dmu"].append(np.abs(means[i] - mu[j, :, 0]).max())
            stats["dcov"].append(np.linalg.norm(covs[i] - cv[j]) / np.linalg.norm(cv[j]))
            stats["dscore"].append(np.abs(scores[i] - s[j]).max())
    frac = stats["matched"] / max(1, max(stats["dev"], stats["ref"]))
    dmu, dcov, dscore = np.asarray(stats["dmu"]), np.asarray(stats["dcov"]), np.asarray(stats["dscore"])
    print("detection-level %s vs float64: %d / %d detections matched (%.3f), kept-set Jaccard %.4f, |dmu| px median %.3g p90 %.3g max %.3g, "
          "|dSigma|/|Sigma| median %.3g p90 %.3g, |dscore| median %.3g p90 %.3g"
          % (precision, stats["matched"], max(stats["dev"], stats["ref"]), frac, float(np.mean(stats["kept_jaccard"])),
             np.median(dmu), np.quantile(dmu, 0.9), dmu.max(), np.median(dcov), np.quantile(dcov, 0.9),
             np.median(dscore), np.quantile(dscore, 0.9)))
    assert stats["ref"] >= 20
    if precision in ("bf16x3", "f16mx", "f16mx4"):   # the parity modes: the same detections, 1e-3
        assert frac == 1.0 and np.mean(stats["kept_jaccard"]) > 0.995
        # (f16mx4: one box of 200 at 0.014 px; the others like f16mx)
        assert dmu.max() < (3e-2 if precision == "f16mx4" else 1e-2) and np.quantile(dcov, 0.9) < 1e-3 and np.quantile(dscore, 0.9) < 1e-3
        return
    assert frac > 0.85
    assert np.mean(stats["kept_jaccard"]) > 0.97
    assert np.median(dmu) < 0.25 and np.quantile(dmu, 0.9) < 1.5          # pixels (boxes are tens of pixels wide)
    assert np.median(dcov) < 0.08
    assert np.median(dscore) < 0.02
