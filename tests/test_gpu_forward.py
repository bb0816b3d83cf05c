"""GPU: whole RetinaNet forward (stem + ResNet-50 + FPN + MC-dropout heads) through the C ABI.

Two different statements are checked (DESIGN.md "Numerics"):

 * Kernel-level parity on identical inputs is the 1e-3 gate and lives in test_gpu_conv.py.
 * End to end, the fast path stores activations as bf16 (BASELINE.json north_star: "bf16 MFMA").
   Two bf16 pipelines that differ only in fp32 summation order decorrelate to the bf16 rounding
   noise floor within a few layers (a 1-ulp flip perturbs 2304 downstream sums), so element-wise
   1e-3 agreement with ANY reference is impossible in this mode.  What must hold instead: the
   device's distance to the float64 ground truth equals the distance of the oracle's own
   bf16-storage emulation to it (same noise floor, nothing added), every tensor is within
   2% relative RMS of float64, and the dropout pattern is bit-identical to the Philox contract.
"""
import numpy as np
import pytest

from conftest import ANCHOR_CFG

pytestmark = pytest.mark.gpu


def _rms(x):
    return float(np.sqrt((np.asarray(x, np.float64) ** 2).mean()))


def _setup(hw, batch, n):
    from bayes_od_rc_amd import synthetic
    from bayes_od_rc_amd.engine import Engine, make_config
    w = synthetic.make_weights()
    frames = synthetic.make_frames(batch, hw[0], hw[1], seed=3)
    eng = Engine(make_config(hw, batch=batch, mc_samples=n))
    eng.load_weights(w)
    return w, frames, eng


_ORACLE_CACHE = {}


def _oracle_key(w, frame, n, seed, image_id, P, depth):
    import hashlib
    probe = w["conv1"]
    while isinstance(probe, (dict, tuple, list)):          # the first array of the first layer: enough to tell two weight sets apart
        probe = next(iter(probe.values())) if isinstance(probe, dict) else probe[0]
    wsum = float(np.asarray(probe, np.float64).sum())
    return (hashlib.sha1(np.ascontiguousarray(frame).tobytes()).hexdigest(), frame.shape, n, seed, image_id, P, depth, len(w), wsum)


def _oracle_f64(w, frame, n, seed, image_id, P, **kw):
    """The float64 literal forward of one frame -- computed once per (weights, frame, masks) in the session: the precision modes of a
    parametrized test compare against the same oracle run (suite time: the NumPy forward of a 192 x 624 frame takes seconds)."""
    from oracle import network, philox
    key = ("f64",) + _oracle_key(w, frame, n, seed, image_id, P, kw.get("backbone_depth", 50))
    if key not in _ORACLE_CACHE:
        km = (lambda s, lid: philox.dropout_keep_mask(seed, image_id, s, lid, P, 256, 0.3)) if n > 1 else None
        _ORACLE_CACHE[key] = network.retinanet_forward(w, frame[None], n, 8, mode="literal", dtype=np.float64, keep_masks=km, return_pyramid=True, **kw)
    return _ORACLE_CACHE[key]


def _oracles(w, frame, n, seed, image_id, P, emulation=True):
    from oracle import network, philox
    emu = None
    if emulation:
        km = (lambda s, lid: philox.dropout_keep_mask(seed, image_id, s, lid, P, 256, 0.3)) if n > 1 else None
        emu = network.retinanet_forward(w, frame[None], n, 8, mode="bf16", keep_masks=km, return_pyramid=True)
    return emu, _oracle_f64(w, frame, n, seed, image_id, P)


@pytest.mark.parametrize("hw,batch,n", [((128, 128), 2, 3), ((96, 160), 1, 2), ((128, 192), 1, 1)])
def test_forward_at_bf16_noise_floor(hw, batch, n):
    seed, first = 1234567890123, 7
    w, frames, eng = _setup(hw, batch, n)
    eng.forward(frames, seed=seed, first_image_id=first)
    cls, box, cov = eng.get_raw()
    pyr = [eng.get_pyramid(l) for l in range(5)]
    for b in range(batch):
        emu, f64 = _oracles(w, frames[b], n, seed, first + b, eng.P)
        items = [("P%d" % (l + 3), pyr[l][b], emu["_pyramid"][l][0], f64["_pyramid"][l][0]) for l in range(5)]
        items += [("cls", cls[b], emu["anchors_class_predictions"], f64["anchors_class_predictions"]),
                  ("box", box[b], emu["anchors_box_predictions"], f64["anchors_box_predictions"]),
                  ("cov", cov[b], emu["_covar_params"], f64["_covar_params"])]
        for name, got, e, t in items:
            assert got.shape == t.shape, name
            d_hip = _rms(got - t) / _rms(t)
            d_emu = _rms(e - t) / _rms(t)
            assert d_hip < 2e-2, (name, d_hip)                     # bf16 storage noise, ~50 layers deep
            assert d_hip < 1.5 * d_emu + 1e-4, (name, d_hip, d_emu)  # no error beyond the emulation's own
        if n > 1:
            # the dropout pattern of the last tower layer is visible as exact zeros in no output,
            # but identical masks imply the heads' sample-to-sample differences correlate ~1 with
            # the oracle's: a wrong mask stream would decorrelate them completely
            dg = (cls[b][0] - cls[b][1]).ravel()
            de = (emu["anchors_class_predictions"][0] - emu["anchors_class_predictions"][1]).ravel()
            assert np.corrcoef(dg, de)[0, 1] > 0.99


REAL_GEOMETRIES = [((720, 1280), "bdd"), ((512, 1696), "kitti")]


@pytest.mark.parametrize("hw,dataset", REAL_GEOMETRIES)
def test_the_reference_s_real_frame_sizes_against_the_cpu_forward(hw, dataset):
    """The sizes the reference actually feeds the network: BDD frames at their native 720 x 1280
    (bdd_dataset_handler.py:128-139: no resize) and KITTI frames resized / padded to 512 x 1696
    (kitti_dataset_handler.py:125-132) -- P = 19 220 / 18 080 pyramid pixels, the 23 -> 45 nearest up-sampling whose ratio is not
    an integer, odd SAME strides, stage-2 rows of 320 / 424 pixels (5 / 7 column strips of the sliding-window kernel, not multiples
    of 64), a P7 level of 6 x 10 / 4 x 14.  One full-size frame, N = 2 MC samples, against oracle/torch_ref.py's fp32 forward with
    the same Philox masks: bf16x3 (the parity mode) element-wise within north_star's 1e-3 on pyramid and head outputs, bf16 (the
    throughput mode) at its storage-noise floor (relative RMS < 2 %, the bound of test_forward_at_bf16_noise_floor, whose
    emulation oracle is too slow at this size) with the MC sample differences correlated > 0.99 with the reference's."""
    from bayes_od_rc_amd import synthetic
    from bayes_od_rc_amd.engine import Engine, make_config
    from oracle import philox, torch_ref
    seed, first, n = 20241003, 11, 2
    w = synthetic.make_weights()
    frame = synthetic.make_frames(1, hw[0], hw[1], seed=5)
    got = {}
    for precision in ("bf16x3", "bf16"):
        eng = Engine(make_config(hw, batch=1, mc_samples=n, precision=precision))
        eng.load_weights(w)
        eng.forward(frame, seed=seed, first_image_id=first)
        cls, box, cov = eng.get_raw()
        got[precision] = {"cls": cls[0].copy(), "box": box[0].copy(), "cov": cov[0].copy(), "pyr": [eng.get_pyramid(l)[0].copy() for l in range(5)]}
        P = eng.P
        levels = eng.levels
        eng.close()
    want_levels = {"bdd": [(90, 160), (45, 80), (23, 40), (12, 20), (6, 10)], "kitti": [(64, 212), (32, 106), (16, 53), (8, 27), (4, 14)]}[dataset]
    assert [tuple(l) for l in levels] == want_levels and P == sum(h * ww for h, ww in want_levels)
    ref = torch_ref.retinanet_forward(w, frame, n, 8, keep_masks=lambda s_, lid: philox.dropout_keep_mask(seed, first, s_, lid, P, 256, 0.3))
    items = lambda g: [("P%d" % (l + 3), g["pyr"][l], ref["_pyramid"][l][0]) for l in range(5)] + [
        ("cls", g["cls"], ref["anchors_class_predictions"]), ("box", g["box"], ref["anchors_box_predictions"]), ("cov", g["cov"], ref["_covar_params"])]
    worst = 0.0
    for name, a, t in items(got["bf16x3"]):
        assert a.shape == t.shape, name
        rms = _rms(t)
        err = float(np.max(np.abs(a - t) / (np.abs(t) + rms)))
        worst = max(worst, err)
        assert err < 1e-3, (name, err)
    for name, a, t in items(got["bf16"]):
        assert _rms(a - t) / _rms(t) < 2e-2, (name, _rms(a - t) / _rms(t))
    dg = (got["bf16"]["cls"][0] - got["bf16"]["cls"][1]).ravel()
    dr = (ref["anchors_class_predictions"][0] - ref["anchors_class_predictions"][1]).ravel()
    assert np.corrcoef(dg, dr)[0, 1] > 0.99
    print("%dx%d: bf16x3 max rel err %.2e vs the fp32 CPU forward" % (hw[0], hw[1], worst))


def test_n1_disables_dropout_and_is_seed_independent():
    """mc_dropout_samples == 1 => dropout off (retinanet_model.py:74-77); BASELINE config 2."""
    w, frames, eng = _setup((128, 128), 1, 1)
    eng.forward(frames, seed=5, first_image_id=0)
    cls, box, cov = eng.get_raw()
    eng.forward(frames, seed=6, first_image_id=3)
    cls2, box2, cov2 = eng.get_raw()
    assert np.array_equal(cls, cls2) and np.array_equal(box, box2) and np.array_equal(cov, cov2)


def test_forward_is_deterministic_and_image_id_keyed():
    w, frames, eng = _setup((128, 128), 2, 2)
    eng.forward(frames, seed=9, first_image_id=4)
    a = eng.get_raw()[0].copy()
    eng.forward(frames, seed=9, first_image_id=4)
    assert np.array_equal(a, eng.get_raw()[0])
    # image 1 of a batch starting at id 4 == image 0 of a batch starting at id 5 (same frame)
    eng.forward(frames[::-1].copy(), seed=9, first_image_id=5)
    b = eng.get_raw()[0]
    assert np.array_equal(a[1], b[0])
    assert not np.array_equal(a[0], b[1])          # same frame, different image id => other masks


def test_geometry_mismatch_is_rejected():
    """An input whose conv pyramid disagrees with the anchor grid (the reference would fail in
    tf.concat) is reported as ValueError, not computed."""
    from bayes_od_rc_amd.engine import Engine, make_config
    with pytest.raises(ValueError):
        Engine(make_config((100, 100), batch=1, mc_samples=2))


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "f16mx", "f16mx4"])
@pytest.mark.parametrize("hw,batch,n,depth", [((128, 128), 2, 3, 50), ((96, 160), 1, 1, 50), ((96, 96), 1, 2, 101)])
def test_fp32_mode_end_to_end(hw, batch, n, depth, precision):
    """precision='fp32' (fp32 storage + exact-fp32 MFMA), precision='bf16x3' ((hi, lo) bf16 pairs, three products on the
    bf16 MFMA: the parity mode of the throughput path) and precision='f16mx' (round 5: bf16x3 with the head towers on one f16 +
    half a block-scaled e2m3 product per multiplication; 'f16mx4': the cross terms as e2m1 products, ~4x the rounding error): the whole forward pass -- stem, 53 backbone
    convs, FPN, MC-dropout heads -- agrees with the float64 oracle element-wise within the 1e-3 bar of
    BASELINE.json's north_star (observed ~1e-5, fp32 summation noise through ~50 layers)."""
    from bayes_od_rc_amd import synthetic
    from bayes_od_rc_amd.engine import Engine, make_config
    seed, first = 99, 3
    w = synthetic.make_weights(depth=depth)           # depth 101: the build's ResNet-101 option (BASELINE config 5)
    frames = synthetic.make_frames(batch, hw[0], hw[1], seed=3)
    eng = Engine(make_config(hw, batch=batch, mc_samples=n, precision=precision, backbone_depth=depth))
    eng.load_weights(w)
    assert eng.plan_info()["tower_mx"] == (precision in ("f16mx", "f16mx4"))      # (the f16mx towers run at every size: the kernel IS the arithmetic)
    eng.forward(frames, seed=seed, first_image_id=first)
    cls, box, cov = eng.get_raw()
    pyr = [eng.get_pyramid(l) for l in range(5)]
    worst = 0.0
    for b in range(batch):
        _, f64 = _oracles(w, frames[b], n, seed, first + b, eng.P, emulation=False)
        items = [("P%d" % (l + 3), pyr[l][b], f64["_pyramid"][l][0]) for l in range(5)]
        items += [("cls", cls[b], f64["anchors_class_predictions"]), ("box", box[b], f64["anchors_box_predictions"]),
                  ("cov", cov[b], f64["_covar_params"])]
        for name, got, t in items:
            rms = _rms(t)
            err = float(np.max(np.abs(got - t) / (np.abs(t) + rms)))
            worst = max(worst, err)
            assert err < 1e-3, (name, err)
            assert _rms(got - t) / rms < (3e-4 if precision == "f16mx4" else 1e-4), (name, _rms(got - t) / rms)
    print("end-to-end forward, precision %s, %dx%d depth %d: max rel err %.2e" % (precision, hw[0], hw[1], depth, worst))


def test_f16mx4_with_the_box_tower_on_hx_rows(monkeypatch):
    """BOD_MX4_BOX_HX=1: the box-regression tower keeps the e2m3 cross terms (hx rows) while the classification and covariance towers
    run on h4 rows -- layer 0 reads ONE hx pyramid and writes each head's format, the two formats are separate launches from layer 1
    on (engine.hip).  Same bound as the pure forms; the box outputs must then be as close as f16mx's."""
    from bayes_od_rc_amd import synthetic
    from bayes_od_rc_amd.engine import Engine, make_config
    hw, batch, n, seed, first = (128, 128), 2, 3, 99, 3
    w = synthetic.make_weights()
    frames = synthetic.make_frames(batch, hw[0], hw[1], seed=3)
    errs = {}
    for mixed in (False, True):
        if mixed:
            monkeypatch.setenv("BOD_MX4_BOX_HX", "1")
        eng = Engine(make_config(hw, batch=batch, mc_samples=n, precision="f16mx4"))
        eng.load_weights(w)
        assert eng.plan_info()["tower_mx_format"] == 2
        eng.forward(frames, seed=seed, first_image_id=first)
        cls, box, cov = eng.get_raw()
        worst = {"cls": 0.0, "box": 0.0, "cov": 0.0}
        for b in range(batch):
            _, f64 = _oracles(w, frames[b], n, seed, first + b, eng.P, emulation=False)
            for name, got, t in (("cls", cls[b], f64["anchors_class_predictions"]), ("box", box[b], f64["anchors_box_predictions"]), ("cov", cov[b], f64["_covar_params"])):
                worst[name] = max(worst[name], float(np.max(np.abs(got - t) / (np.abs(t) + _rms(t)))))
        errs[mixed] = worst
        eng.close()
    print("f16mx4, all towers on h4: %s; box tower on hx: %s" % (errs[False], errs[True]))
    assert all(v < 1e-3 for v in errs[False].values()) and all(v < 1e-3 for v in errs[True].values())
    assert errs[True]["box"] < 0.6 * errs[False]["box"]          # (e2m3 cross terms: ~4x less rounding error than e2m1)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "f16mx", "f16mx4"])
@pytest.mark.parametrize("hw", [(192, 624), (360, 640)])
def test_non_square_and_odd_pyramids(hw, precision):
    """Half-scale versions of BASELINE config 4 (KITTI 384x1248 -> 192x624: odd level widths 78/39/20/10/5)
    and of the real BDD frame (720x1280 -> 360x640: 45->23->12->6->3, nearest up-sampling with a
    non-integer ratio, stride-2 SAME on odd sizes).  fp32 mode against the float64 oracle, 1e-3."""
    from bayes_od_rc_amd import synthetic
    from bayes_od_rc_amd.engine import Engine, make_config
    n, seed = 2, 17
    w = synthetic.make_weights()
    frames = synthetic.make_frames(1, hw[0], hw[1], seed=9)
    eng = Engine(make_config(hw, batch=1, mc_samples=n, precision=precision))
    eng.load_weights(w)
    eng.forward(frames, seed=seed, first_image_id=0)
    cls, box, cov = eng.get_raw()
    _, f64 = _oracles(w, frames[0], n, seed, 0, eng.P, emulation=False)
    assert [tuple(p.shape[1:3]) for p in f64["_pyramid"]] == eng.levels
    for l in range(5):
        t = f64["_pyramid"][l][0]
        assert float(np.max(np.abs(eng.get_pyramid(l)[0] - t) / (np.abs(t) + _rms(t)))) < 1e-3, l
    for got, key in ((cls[0], "anchors_class_predictions"), (box[0], "anchors_box_predictions"), (cov[0], "_covar_params")):
        t = f64[key]
        assert float(np.max(np.abs(got - t) / (np.abs(t) + _rms(t)))) < 1e-3, key
    if precision != "fp32":
        return
    # and the bf16 path stays at its noise floor on the same geometry
    eng16 = Engine(make_config(hw, batch=1, mc_samples=n))
    eng16.load_weights(w)
    eng16.forward(frames, seed=seed, first_image_id=0)
    c16 = eng16.get_raw()[0][0]
    assert _rms(c16 - f64["anchors_class_predictions"]) / _rms(f64["anchors_class_predictions"]) < 2e-2


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_fused_head_output_equals_separate_launches(precision):
    """The 1x1 head output convs fused into the last tower layer's epilogue (256-wide cout tile) give the same raw head
    outputs as the separate 1x1 launches: to fp32 round-off in the bf16 mode (the fused form multiplies the bf16 tile in LDS,
    the separate launch the same values from HBM, in another summation order), BIT FOR BIT in the bf16x3 mode, whose fused form
    issues the (hi, lo) products of the separate launch in the same order per accumulator."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from bayes_od_rc_amd import synthetic\n"
            "from bayes_od_rc_amd.engine import Engine, make_config\n"
            "eng = Engine(make_config((128, 160), batch=2, mc_samples=3, precision=%r))\n"
            "eng.load_weights(synthetic.make_weights())\n"
            "eng.profile_begin(which=0)\n"
            "eng.forward(synthetic.make_frames(2, 128, 160, seed=6), seed=21, first_image_id=4)\n"
            "print('HEAD_LAUNCHES', eng.profile_end()['head_conv_launches'])\n"
            "c, b, v = eng.get_raw()\n"
            "np.savez(sys.argv[1], c=c, b=b, v=v)\n" % (root, precision))
    outs = []
    for fuse in ("1", "0"):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "o.npz")
            env = dict(os.environ, BOD_FORCE_CONV_TILE="256", BOD_FUSE_HEAD_OUTPUT=fuse)
            r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-3000:]
            # the tower launches (1x1 launches are not head 3x3 launches): four, or five on the plan with the fused MC aggregation, whose
            # layer 2 runs the ending regression head (sample-complete tiles) and the two continuing heads (plain tiles) separately
            assert ("HEAD_LAUNCHES 5" if fuse == "1" else "HEAD_LAUNCHES 4") in r.stdout, r.stdout[-300:]
            z = np.load(path)
            outs.append({k: z[k] for k in z.files})
    for k in ("c", "b", "v"):
        a, b = outs[0][k], outs[1][k]
        assert a.shape == b.shape and np.abs(b).max() > 0
        if precision == "bf16x3":
            assert np.array_equal(a, b), k
        else:
            assert np.max(np.abs(a - b)) <= 1e-5 * max(1.0, float(np.abs(b).max())), k


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_activation_row_reuse_is_bit_identical(precision):
    """Row-reuse staging of the 3x3 tower layers (extended rows shared by the three kx taps) changes
    only how activations reach LDS, not a single bit of the result -- in the bf16 mode (although the row-reuse kernel multiplies
    with the 16x16x32 MFMA shape and the generic loop with 32x32x16) and in the bf16x3 mode (whose row-reuse loop issues the
    same (hi, lo) products in the same order per accumulator as the generic loop)."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from bayes_od_rc_amd import synthetic\n"
            "from bayes_od_rc_amd.engine import Engine, make_config\n"
            "eng = Engine(make_config((96, 160), batch=2, mc_samples=3, precision=%r))\n"
            "eng.load_weights(synthetic.make_weights())\n"
            "eng.forward(synthetic.make_frames(2, 96, 160, seed=8), seed=5, first_image_id=1)\n"
            "c, b, v = eng.get_raw()\n"
            "np.savez(sys.argv[1], c=c, b=b, v=v)\n" % (root, precision))
    outs = []
    for xr in ("1", "0"):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "o.npz")
            env = dict(os.environ, BOD_FORCE_CONV_TILE="256", BOD_CONV_XREUSE=xr)
            subprocess.run([sys.executable, "-c", code, path], check=True, env=env)
            z = np.load(path)
            outs.append({k: z[k] for k in z.files})
    # (bf16: the row-reuse kernel multiplies with v_mfma_f32_16x16x32_bf16, the generic loop with 32x32x16 -- the matrix pipe
    # accumulates both in the same order over k, so even that changes no bit)
    for k in ("c", "b", "v"):
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_mid_tile_barrier_tower_loop_is_bit_identical():
    """The per-sample tower launches run on the loop whose K-tile barrier sits two fragment steps before the K-tile's end, with
    inline-asm fragment reads and hand-counted lgkmcnt waits (conv_igemm.hip); BOD_TOWER_MIDBAR=0 selects the round-2 loop
    (barrier at the top of the K-tile, compiler-placed waits).  Same products in the same
    order per accumulator: raw head outputs of a forward AND the detections of an aggregating infer must not change by one bit --
    any fragment consumed before it landed, or a stage overwritten while it is read, shows up here."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, sys; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')\n"
            "from conftest import ANCHOR_CFG, BAYES_CFG, NMS_CFG\n"
            "from bayes_od_rc_amd import synthetic\n"
            "from bayes_od_rc_amd.anchor_generator import FpnAnchorGenerator\n"
            "from bayes_od_rc_amd.engine import Engine, make_config\n"
            "out = {}\n"
            "for hw, b, n in (((96, 160), 2, 3), ((160, 160), 3, 10)):\n"
            "    eng = Engine(make_config(hw, batch=b, mc_samples=n, bayes_od_config=BAYES_CFG, nms_config=NMS_CFG, use_full_covar=True))\n"
            "    eng.load_weights(synthetic.make_weights(cls_fg_bias=-1.0))\n"
            "    eng.set_anchors(FpnAnchorGenerator(ANCHOR_CFG).generate_all((hw[0], hw[1], 3)))\n"
            "    fr = synthetic.make_frames(b, hw[0], hw[1], seed=8)\n"
            "    for rep in range(3):\n"
            "        eng.infer(fr, seed=5 + rep, first_image_id=rep)\n"
            "        for k, v in zip('smcn', eng.get_detections(b - 1)): out['det%%d_%%d_%%s' %% (n, rep, k)] = v\n"
            "        for k, v in eng.get_posterior(0).items(): out['post%%d_%%d_%%s' %% (n, rep, k)] = v\n"
            "    eng.forward(fr, seed=5, first_image_id=1)\n"
            "    for k, v in zip('cbv', eng.get_raw()): out['raw%%d_%%s' %% (n, k)] = v\n"
            "    eng.close()\n"
            "np.savez(sys.argv[1], **out)\n" % (root, root))
    outs = []
    for mb in ("1", "0"):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "o.npz")
            env = dict(os.environ, BOD_FORCE_CONV_TILE="256", BOD_TOWER_MIDBAR=mb)
            r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-3000:]
            z = np.load(path)
            outs.append({k: z[k] for k in z.files})
    assert set(outs[0]) == set(outs[1]) and len(outs[0]) > 20
    for k in sorted(outs[0]):
        assert outs[0][k].shape == outs[1][k].shape, k
        assert np.array_equal(outs[0][k], outs[1][k]), k
    assert np.abs(outs[0]["raw10_c"]).max() > 0


def test_bottleneck_chain_fusion_is_bit_identical():
    """Stages 2 and 3 of the backbone run each block's 3x3 conv with the block's 1x1 expansion (+ shortcut + ReLU) and the next
    block's 1x1 reduction fused onto its LDS tile (conv_igemm.hip, ABL = 10: same MFMA shape, k order and epilogue arithmetic as the
    separate launches).  BOD_CHAIN_FUSION=0 plans the separate launches: the FPN pyramid and the raw head outputs must not differ
    by one bit, for ResNet-50 and -101, square and non-square frames, batch 1 (where stage-3 layers go split-K and stay unfused)
    and batch 3."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from bayes_od_rc_amd import synthetic\n"
            "from bayes_od_rc_amd.engine import Engine, make_config\n"
            "out = {}\n"
            "for tag, hw, b, depth in (('a', (128, 128), 3, 50), ('b', (96, 160), 1, 50), ('c', (192, 624), 2, 50), ('d', (128, 128), 2, 101)):\n"
            "    eng = Engine(make_config(hw, batch=b, mc_samples=2, backbone_depth=depth))\n"
            "    eng.load_weights(synthetic.make_weights(depth=depth))\n"
            "    eng.forward(synthetic.make_frames(b, hw[0], hw[1], seed=3), seed=11, first_image_id=2)\n"
            "    for l in range(5): out['%%s_p%%d' %% (tag, l)] = eng.get_pyramid(l)\n"
            "    for k, v in zip('cbv', eng.get_raw()): out['%%s_%%s' %% (tag, k)] = v\n"
            "    out[tag + '_ops'] = np.int32(eng.plan_info()['ops'])\n"
            "    eng.close()\n"
            "np.savez(sys.argv[1], **out)\n" % root)
    outs = []
    for fuse in ("3", "0"):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "o.npz")
            r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, BOD_CHAIN_FUSION=fuse), capture_output=True, text=True)      # 3: stages 2 AND 3 chained
            assert r.returncode == 0, r.stderr[-3000:]
            z = np.load(path)
            outs.append({k: z[k] for k in z.files})
    fused, plain = outs
    # the fused plan really is shorter: 2 launches fewer per chained block (7 blocks in stages 2-3; batch 1 keeps some split-K layers apart)
    assert int(fused["c_ops"]) < int(plain["c_ops"]) and int(fused["a_ops"]) < int(plain["a_ops"])
    for k in sorted(fused):
        if k.endswith("_ops"):
            continue
        assert fused[k].shape == plain[k].shape and np.abs(plain[k]).max() > 0, k
        assert np.array_equal(fused[k], plain[k]), k


def test_row_task_stem_kernel_is_bit_identical():
    """The bf16-mode stem runs 256-pixel row tasks with the weights in registers since round 3 (aux_kernels.hip,
    stem_conv_bf16_row_kernel: same MFMAs in the same k order per accumulator, hardware round-to-nearest-even packs).
    BOD_STEM_SEG64=1 runs the 64-pixel-task kernel it replaces: the pyramid must not differ by one bit -- square frames, a
    row of more than one 256-pixel segment with a ragged tail (624 -> 309 outputs), and a width that is not a multiple of
    four (the launcher then keeps the old kernel in both runs)."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from bayes_od_rc_amd import synthetic\n"
            "from bayes_od_rc_amd.engine import Engine, make_config\n"
            "out = {}\n"
            "for tag, hw, b in (('a', (128, 128), 3), ('b', (192, 624), 2), ('c', (96, 1160), 1), ('d', (128, 126), 1)):\n"
            "    eng = Engine(make_config(hw, batch=b, mc_samples=1))\n"
            "    eng.load_weights(synthetic.make_weights())\n"
            "    eng.forward(synthetic.make_frames(b, hw[0], hw[1], seed=4), seed=1, first_image_id=0)\n"
            "    for l in range(5): out['%%s_p%%d' %% (tag, l)] = eng.get_pyramid(l)\n"
            "    eng.close()\n"
            "np.savez(sys.argv[1], **out)\n" % root)
    outs = []
    for old in ("0", "1"):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "o.npz")
            r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, BOD_STEM_SEG64=old), capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-3000:]
            z = np.load(path)
            outs.append({k: z[k] for k in z.files})
    assert set(outs[0]) == set(outs[1]) and len(outs[0]) == 20
    for k in sorted(outs[0]):
        assert np.array_equal(outs[0][k], outs[1][k]), k
        assert np.isfinite(outs[0][k]).all() and np.abs(outs[0][k]).max() > 0, k


def test_pointwise_kernel_is_bit_identical():
    """1x1 layers that reduce at most 256 channels run on the streaming pointwise kernel since round 3 (conv_pointwise.hip:
    persistent workgroups, weights in registers, the next tile's rows in flight; same MFMA shape, k order and epilogue
    arithmetic as the generic kernel).  BOD_POINTWISE=0 plans the generic launches: pyramid and raw head outputs must not differ
    by one bit -- all three reduction widths (64 / 128 / 256), with and without shortcut, stride-2 layers, ragged last tiles,
    ResNet-50 and -101 (BOD_POINTWISE_MIN_M=1 sends small test shapes down the pointwise path too)."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from bayes_od_rc_amd import synthetic\n"
            "from bayes_od_rc_amd.engine import Engine, make_config\n"
            "out = {}\n"
            "for tag, hw, b, depth in (('a', (128, 128), 3, 50), ('b', (96, 160), 1, 50), ('c', (192, 624), 2, 50), ('d', (128, 128), 2, 101), ('e', (256, 256), 9, 50)):\n"
            "    eng = Engine(make_config(hw, batch=b, mc_samples=2, backbone_depth=depth))\n"
            "    eng.load_weights(synthetic.make_weights(depth=depth))\n"
            "    eng.forward(synthetic.make_frames(b, hw[0], hw[1], seed=3), seed=11, first_image_id=2)\n"
            "    for l in range(5): out['%%s_p%%d' %% (tag, l)] = eng.get_pyramid(l)\n"
            "    for k, v in zip('cbv', eng.get_raw()): out['%%s_%%s' %% (tag, k)] = v\n"
            "    eng.close()\n"
            "np.savez(sys.argv[1], **out)\n" % root)
    outs = []
    for on in ("1", "0"):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "o.npz")
            env = dict(os.environ, BOD_POINTWISE=on, BOD_POINTWISE_MIN_M="1", BOD_CONV_SPLITK="0")
            r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-3000:]
            z = np.load(path)
            outs.append({k: z[k] for k in z.files})
    assert set(outs[0]) == set(outs[1]) and len(outs[0]) == 40
    for k in sorted(outs[0]):
        assert np.array_equal(outs[0][k], outs[1][k]), (k, float(np.abs(outs[0][k].astype(np.float64) - outs[1][k]).max()))
        assert np.isfinite(outs[0][k]).all(), k


def test_sliding_window_3x3_128_channel_kernel_matches_the_generic_launches():
    """ResNet stage 3's 3x3 layers (128 -> 128 channels) run on their own sliding-window kernel since round 4 (conv_pointwise.hip:
    eight waves = 4 cout blocks x 2 halves of the input channels, partial sums traded through LDS).  Its sum per pixel is the generic
    kernel's with ONE fp32 addition re-associated ((channels 0-63) + (channels 64-127)), so the comparison with BOD_SLIDE3X3_C128=0 is
    not bit-exact: a layer's bf16 outputs flip by one ulp here and there (tests/test_gpu_conv.py pins exactly that on single layers),
    and thirty layers further down the two pyramids are two bf16 roundings of the same network -- every level within 1e-2 relative
    RMS of the other, as far as either is from the oracle (test_forward_at_bf16_noise_floor) -- widths that are and are not multiples
    of 64, one and several strips per row, ResNet-50 and -101.  The two plans must really differ."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from bayes_od_rc_amd import synthetic\n"
            "from bayes_od_rc_amd.engine import Engine, make_config\n"
            "out = {}\n"
            "for tag, hw, b, depth in (('a', (128, 128), 3, 50), ('b', (96, 160), 1, 50), ('c', (192, 624), 2, 50), ('d', (128, 128), 2, 101), ('e', (256, 256), 9, 50), ('f', (512, 1040), 2, 50)):\n"
            "    eng = Engine(make_config(hw, batch=b, mc_samples=2, backbone_depth=depth))\n"
            "    eng.load_weights(synthetic.make_weights(depth=depth))\n"
            "    eng.forward(synthetic.make_frames(b, hw[0], hw[1], seed=3), seed=11, first_image_id=2)\n"
            "    for l in range(5): out['%%s_p%%d' %% (tag, l)] = eng.get_pyramid(l)\n"
            "    eng.close()\n"
            "np.savez(sys.argv[1], **out)\n" % root)
    outs = []
    for on in ("1", "0"):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "o.npz")
            env = dict(os.environ, BOD_SLIDE3X3_C128=on, BOD_POINTWISE_MIN_M="1", BOD_CONV_SPLITK="0")
            r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-3000:]
            z = np.load(path)
            outs.append({k: z[k] for k in z.files})
    assert set(outs[0]) == set(outs[1]) and len(outs[0]) == 30
    differs = 0
    for k in sorted(outs[0]):
        a, b = outs[0][k].astype(np.float64), outs[1][k].astype(np.float64)
        assert np.isfinite(a).all() and np.abs(a).max() > 0, k
        rel = np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean())
        assert rel < 1e-2, (k, rel)
        differs += int(not np.array_equal(a, b))
    assert differs > 0, "BOD_SLIDE3X3_C128 did not change the plan"


def test_cout_inner_grid_order_is_bit_identical():
    """Round 4: launches of the generic kernel with several cout tiles per pixel tile (ResNet stage 5: 2 and 8 of them) are one grid
    row with the cout tile as the fast index inside an XCD from 64 pixel tiles on (conv_igemm.hip; BOD_COUT_INNER=0: the (nx, ny)
    grid).  Only WHERE a tile runs changes: 64 frames of 512 x 512 (64 pixel tiles in stage 5, 1 024 in stage 3's projection) and a
    ragged 70-frame batch (XCDs with one pixel tile less retire their last slots at once) must give the same pyramid, bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, sys, json; sys.path.insert(0, %r)\n"
            "from bayes_od_rc_amd import synthetic\n"
            "from bayes_od_rc_amd.engine import Engine, make_config\n"
            "out = []\n"
            "for b in (64, 70):\n"
            "    eng = Engine(make_config((512, 512), batch=b, mc_samples=1))\n"
            "    eng.load_weights(synthetic.make_weights())\n"
            "    eng.forward(synthetic.make_frames(b, 512, 512, seed=4), seed=1, first_image_id=0)\n"
            "    for l in range(5):\n"
            "        v = np.ascontiguousarray(eng.get_pyramid(l)).view(np.uint32).ravel()\n"
            "        out.append([int(v.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(v))])\n"
            "    eng.close()\n"
            "print('CHECKSUMS ' + json.dumps(out))\n" % root)
    sums = []
    for on in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BOD_COUT_INNER=on), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        import json
        sums.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("CHECKSUMS ")][0][len("CHECKSUMS "):]))
    assert sums[0] == sums[1] and len(sums[0]) == 10 and all(s[0] != 0 for s in sums[0])


def test_plane_row_reuse_layers_are_bit_identical():
    """Round 4: the plane -> plane 3x3 stride-1 layers of 256 -> 256 channels (stage 4's `2b`, P3-P5) run on the tower kernel's
    row-reuse loop once their launch fills the chip with 256x256 tiles (engine.hip add_conv; BOD_PLANE_XREUSE=0: the generic loop).
    Row reuse changes how activations reach LDS, not the order of the products per accumulator: every pyramid level must not differ
    by one bit -- widths that do and do not fill a tile's runs, ragged last tiles, ResNet-101's 23 stage-4 blocks; one shape at its
    natural tile size (P3 of 24 frames of 512x512 = 384 tiles) without BOD_FORCE_CONV_TILE.  The plan must really differ
    (plan_info()['plane_row_reuse_layers'])."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, sys, os; sys.path.insert(0, %r)\n"
            "from bayes_od_rc_amd import synthetic\n"
            "from bayes_od_rc_amd.engine import Engine, make_config\n"
            "out = {}\n"
            "shapes = (('n', (512, 512), 24, 50),) if os.environ.get('BOD_FORCE_CONV_TILE') is None else "
            "(('a', (128, 128), 3, 50), ('b', (96, 160), 1, 50), ('c', (192, 624), 2, 50), ('d', (128, 128), 2, 101), ('e', (256, 256), 9, 50))\n"
            "for tag, hw, b, depth in shapes:\n"
            "    eng = Engine(make_config(hw, batch=b, mc_samples=2, backbone_depth=depth))\n"
            "    eng.load_weights(synthetic.make_weights(depth=depth))\n"
            "    out['%%s_layers' %% tag] = np.array([eng.plan_info()['plane_row_reuse_layers']])\n"
            "    eng.forward(synthetic.make_frames(b, hw[0], hw[1], seed=3), seed=11, first_image_id=2)\n"
            "    for l in range(5): out['%%s_p%%d' %% (tag, l)] = eng.get_pyramid(l)\n"
            "    eng.close()\n"
            "np.savez(sys.argv[1], **out)\n" % root)
    for forced in ("256", None):
        outs = []
        for on in ("1", "0"):
            with tempfile.TemporaryDirectory() as d:
                path = os.path.join(d, "o.npz")
                env = dict(os.environ, BOD_PLANE_XREUSE=on, BOD_CONV_SPLITK="0")
                env.pop("BOD_FORCE_CONV_TILE", None)
                if forced:
                    env["BOD_FORCE_CONV_TILE"] = forced
                r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True)
                assert r.returncode == 0, r.stderr[-3000:]
                z = np.load(path)
                outs.append({k: z[k] for k in z.files})
        assert set(outs[0]) == set(outs[1]) and len(outs[0]) == (30 if forced else 6)
        for k in sorted(outs[0]):
            if k.endswith("_layers"):
                # forced tiles: 6 (23 for ResNet-101) stage-4 `2b` layers + P3, P4, P5; natural size at 24 frames: P3 only
                want = (26 if k.startswith("d") else 9) if forced else 1
                assert int(outs[0][k][0]) == want and int(outs[1][k][0]) == 0, (k, outs[0][k], outs[1][k])
                continue
            assert np.isfinite(outs[0][k]).all() and np.abs(outs[0][k]).max() > 0, k
            assert np.array_equal(outs[0][k], outs[1][k]), (k, float(np.abs(outs[0][k].astype(np.float64) - outs[1][k]).max()))


def test_sliding_window_3x3_kernel_is_bit_identical():
    """ResNet stage 2's 3x3 layers (64 -> 64 channels) run on the sliding-window kernel (conv_pointwise.hip: a workgroup walks
    down a 64-pixel column strip, three input rows in an LDS ring, every input pixel staged once; same MFMA shape, k order and
    epilogue as the generic kernel).  BOD_SLIDE3X3=0 plans the generic launches: pyramid and raw head outputs must not differ by
    one bit -- widths that are and are not multiples of 64, one and several strips per row, ResNet-50 and -101."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from bayes_od_rc_amd import synthetic\n"
            "from bayes_od_rc_amd.engine import Engine, make_config\n"
            "out = {}\n"
            "for tag, hw, b, depth in (('a', (128, 128), 3, 50), ('b', (96, 160), 1, 50), ('c', (192, 624), 2, 50), ('d', (128, 128), 2, 101), ('e', (256, 256), 9, 50)):\n"
            "    eng = Engine(make_config(hw, batch=b, mc_samples=2, backbone_depth=depth))\n"
            "    eng.load_weights(synthetic.make_weights(depth=depth))\n"
            "    eng.forward(synthetic.make_frames(b, hw[0], hw[1], seed=3), seed=11, first_image_id=2)\n"
            "    for l in range(5): out['%%s_p%%d' %% (tag, l)] = eng.get_pyramid(l)\n"
            "    for k, v in zip('cbv', eng.get_raw()): out['%%s_%%s' %% (tag, k)] = v\n"
            "    eng.close()\n"
            "np.savez(sys.argv[1], **out)\n" % root)
    outs = []
    for on in ("1", "0"):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "o.npz")
            # (stage 3's 128-channel sliding window is not bit-identical by construction: off here, tested on its own)
            env = dict(os.environ, BOD_SLIDE3X3=on, BOD_SLIDE3X3_C128="0", BOD_POINTWISE_MIN_M="1", BOD_CHAIN_FUSION="0")
            r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-3000:]
            z = np.load(path)
            outs.append({k: z[k] for k in z.files})
    assert set(outs[0]) == set(outs[1]) and len(outs[0]) == 40
    for k in sorted(outs[0]):
        assert np.array_equal(outs[0][k], outs[1][k]), (k, float(np.abs(outs[0][k].astype(np.float64) - outs[1][k]).max()))
        assert np.isfinite(outs[0][k]).all(), k


def test_pointwise_fused_next_reduction_is_bit_identical():
    """ResNet stage 2 on the pointwise kernel's 256-channel tile: the 1x1 expansion of a block carries the NEXT block's 1x1
    reduction, computed from the finished tile in LDS (conv_pointwise.hip, NEXT).  BOD_PW_FUSE_NEXT=0 plans the separate
    launches: pyramid and raw head outputs must not differ by one bit; the fused plan is two launches shorter (ResNet-50)."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from bayes_od_rc_amd import synthetic\n"
            "from bayes_od_rc_amd.engine import Engine, make_config\n"
            "out = {}\n"
            "for tag, hw, b, depth in (('a', (128, 128), 3, 50), ('b', (96, 160), 1, 50), ('c', (192, 624), 2, 50), ('d', (128, 128), 2, 101), ('e', (256, 256), 9, 50)):\n"
            "    eng = Engine(make_config(hw, batch=b, mc_samples=2, backbone_depth=depth))\n"
            "    eng.load_weights(synthetic.make_weights(depth=depth))\n"
            "    eng.forward(synthetic.make_frames(b, hw[0], hw[1], seed=3), seed=11, first_image_id=2)\n"
            "    for l in range(5): out['%%s_p%%d' %% (tag, l)] = eng.get_pyramid(l)\n"
            "    for k, v in zip('cbv', eng.get_raw()): out['%%s_%%s' %% (tag, k)] = v\n"
            "    out[tag + '_ops'] = np.int32(eng.plan_info()['ops'])\n"
            "    eng.close()\n"
            "np.savez(sys.argv[1], **out)\n" % root)
    outs = []
    for on in ("1", "0"):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "o.npz")
            env = dict(os.environ, BOD_PW_FUSE_NEXT=on, BOD_POINTWISE="1", BOD_CHAIN_FUSION="0", BOD_POINTWISE_MIN_M="1", BOD_CONV_SPLITK="0")
            r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-3000:]
            z = np.load(path)
            outs.append({k: z[k] for k in z.files})
    assert set(outs[0]) == set(outs[1])
    for k in sorted(outs[0]):
        if k.endswith("_ops"):
            assert int(outs[0][k]) == int(outs[1][k]) - 2, (k, int(outs[0][k]), int(outs[1][k]))
            continue
        assert np.array_equal(outs[0][k], outs[1][k]), (k, float(np.abs(outs[0][k].astype(np.float64) - outs[1][k]).max()))
        assert np.isfinite(outs[0][k]).all(), k


def test_pointwise_dual_convblock_launch_is_bit_identical():
    """Stage 2's ConvBlock on the pointwise kernel's dual form (round 4): the projection shortcut `branch1` (64 -> 256) carries the
    block's own 1x1 reduction `2a` (64 -> 64 + ReLU) on the SAME input tile (conv_pointwise.hip, NEXT = 2; feature_extractor.py:
    283-309).  BOD_PW_FUSE_DUAL=0 plans the two launches in the reference's order: pyramid and raw head outputs must not differ by
    one bit; the fused plan is one launch shorter.  Square / non-square frames, ragged last tiles, ResNet-50 and -101."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from bayes_od_rc_amd import synthetic\n"
            "from bayes_od_rc_amd.engine import Engine, make_config\n"
            "out = {}\n"
            "for tag, hw, b, depth in (('a', (128, 128), 3, 50), ('b', (96, 160), 1, 50), ('c', (192, 624), 2, 50), ('d', (128, 128), 2, 101), ('e', (256, 256), 9, 50), ('f', (360, 640), 2, 50)):\n"
            "    eng = Engine(make_config(hw, batch=b, mc_samples=2, backbone_depth=depth))\n"
            "    eng.load_weights(synthetic.make_weights(depth=depth))\n"
            "    eng.forward(synthetic.make_frames(b, hw[0], hw[1], seed=3), seed=11, first_image_id=2)\n"
            "    for l in range(5): out['%%s_p%%d' %% (tag, l)] = eng.get_pyramid(l)\n"
            "    for k, v in zip('cbv', eng.get_raw()): out['%%s_%%s' %% (tag, k)] = v\n"
            "    out[tag + '_ops'] = np.int32(eng.plan_info()['ops'])\n"
            "    eng.close()\n"
            "np.savez(sys.argv[1], **out)\n" % root)
    outs = []
    for on in ("1", "0"):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "o.npz")
            env = dict(os.environ, BOD_PW_FUSE_DUAL=on, BOD_POINTWISE="1", BOD_CHAIN_FUSION="0", BOD_POINTWISE_MIN_M="1", BOD_CONV_SPLITK="0")
            r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-3000:]
            z = np.load(path)
            outs.append({k: z[k] for k in z.files})
    assert set(outs[0]) == set(outs[1])
    for k in sorted(outs[0]):
        if k.endswith("_ops"):
            assert int(outs[0][k]) == int(outs[1][k]) - 1, (k, int(outs[0][k]), int(outs[1][k]))
            continue
        assert np.array_equal(outs[0][k], outs[1][k]), (k, float(np.abs(outs[0][k].astype(np.float64) - outs[1][k]).max()))
        assert np.isfinite(outs[0][k]).all(), k


def test_fused_stem_pool_is_bit_identical():
    """bf16 inference on stem rows of at most 256 pixels runs stem + ZeroPadding2D((1,2)) + max-pool as ONE kernel (aux_kernels.hip,
    stem_pool_fused_kernel: a workgroup walks down an image with an 8-row input ring, stem rows are max-combined in registers,
    only the pooled rows are stored).  BOD_STEM_POOL_FUSED=0 runs the two launches: the pyramid must not differ by one bit --
    even / odd stem heights, widths below and at the 256-pixel limit, several images."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from bayes_od_rc_amd import synthetic\n"
            "from bayes_od_rc_amd.engine import Engine, make_config\n"
            "out = {}\n"
            "for tag, hw, b in (('a', (128, 128), 3), ('b', (96, 160), 2), ('c', (512, 512), 2), ('d', (192, 512), 1), ('e', (128, 126), 1)):\n"
            "    eng = Engine(make_config(hw, batch=b, mc_samples=1))\n"
            "    eng.load_weights(synthetic.make_weights())\n"
            "    eng.forward(synthetic.make_frames(b, hw[0], hw[1], seed=4), seed=1, first_image_id=0)\n"
            "    for l in range(5): out['%%s_p%%d' %% (tag, l)] = eng.get_pyramid(l)\n"
            "    eng.close()\n"
            "np.savez(sys.argv[1], **out)\n" % root)
    outs = []
    for on in ("1", "0"):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "o.npz")
            env = dict(os.environ, BOD_STEM_POOL_FUSED=on, BOD_STEM_POOL_FUSED_MIN_B="1")
            r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-3000:]
            z = np.load(path)
            outs.append({k: z[k] for k in z.files})
    assert set(outs[0]) == set(outs[1]) and len(outs[0]) == 25
    for k in sorted(outs[0]):
        assert np.array_equal(outs[0][k], outs[1][k]), (k, float(np.abs(outs[0][k].astype(np.float64) - outs[1][k]).max()))
        assert np.isfinite(outs[0][k]).all() and np.abs(outs[0][k]).max() > 0, k


@pytest.mark.parametrize("precision", ["bf16x3", "f16mx"])
def test_fused_stem_pool_of_the_pair_precisions_matches_the_fp32_stem(precision):
    """Round 6: the (hi, lo) precisions run stem + zero-pad + max-pool as one kernel too (aux_kernels.hip,
    stem_pool_fused_split_kernel: three bf16 products like every other conv of those modes, fp32 pooling, pooled pixels stored as
    pairs).  BOD_STEM_SPLIT_FUSED=0 runs the exact fp32 stem + the pooling launch they had before.  The two are not bit-identical
    (three bf16 products carry 4e-6 of the output RMS per layer): every pyramid level must agree within 2e-5 of its RMS in the RMS
    and 2e-4 at the worst element (measured: 6.3e-5) -- the class of the backbone's other convs in these modes, a tenth of what the
    precision-mode tests allow the whole forward -- on even / odd stem heights, widths below and at the 256-pixel limit, a ragged
    last column, several images.  (A wrong tap, border or slot would show at 1e-2 and above.)"""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from bayes_od_rc_amd import synthetic\n"
            "from bayes_od_rc_amd.engine import Engine, make_config\n"
            "out = {}\n"
            "for tag, hw, b in (('a', (128, 128), 3), ('b', (96, 160), 2), ('c', (512, 512), 2), ('d', (192, 512), 1), ('e', (128, 126), 1)):\n"
            "    eng = Engine(make_config(hw, batch=b, mc_samples=1, precision=%r))\n"
            "    eng.load_weights(synthetic.make_weights())\n"
            "    eng.forward(synthetic.make_frames(b, hw[0], hw[1], seed=4), seed=1, first_image_id=0)\n"
            "    for l in range(5): out['%%s_p%%d' %% (tag, l)] = eng.get_pyramid(l)\n"
            "    eng.close()\n"
            "np.savez(sys.argv[1], **out)\n" % (root, precision))
    outs = []
    for on in ("1", "0"):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "o.npz")
            env = dict(os.environ, BOD_STEM_SPLIT_FUSED=on, BOD_STEM_POOL_FUSED_MIN_B="1")
            r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-3000:]
            z = np.load(path)
            outs.append({k: z[k] for k in z.files})
    assert set(outs[0]) == set(outs[1]) and len(outs[0]) == 25
    worst = worst_rms = 0.0
    differs = 0
    for k in sorted(outs[0]):
        a, b = outs[0][k].astype(np.float64), outs[1][k].astype(np.float64)
        assert np.isfinite(a).all() and np.abs(a).max() > 0, k
        rms = np.sqrt(np.mean(b * b))
        rel, rel_rms = float(np.abs(a - b).max() / rms), float(np.sqrt(np.mean((a - b) ** 2)) / rms)
        worst, worst_rms = max(worst, rel), max(worst_rms, rel_rms)
        differs += int(not np.array_equal(a, b))
        assert rel <= 2e-4 and rel_rms <= 2e-5, (k, rel, rel_rms)
    assert differs > 0, "the fused kernel did not run (identical bits: both runs took the fp32 stem)"
    print("fused (hi, lo) stem + pool against the fp32 stem, %s: over 25 pyramid levels max |d| / rms %.2e, rms(d) / rms %.2e" % (precision, worst, worst_rms))


@pytest.mark.parametrize("hw,batch", [((512, 512), 128), ((720, 1280), 12), ((512, 1696), 12)])
def test_streaming_backbone_kernels_are_bit_identical_at_full_size(hw, batch):
    """The round-3 backbone kernels (fused stem + pool, streaming pointwise with stage 2's fused reduction, sliding-window 3x3)
    against the generic launches at the bench's frame size and a batch that fills the chip (128 frames of 512 x 512: two
    workgroups per CU in every persistent kernel, the shapes the explicit vmcnt waits and the LDS rings are exercised hardest
    by), three forwards each: checksums of every pyramid level and of the raw head outputs must agree exactly.  Round 4: also at
    the reference's real frame sizes -- 720 x 1280 (stage-2 rows of 320 pixels: 5 column strips) and 512 x 1696 (424
    pixels: 7 strips, the last one 40 pixels wide; ragged pointwise tiles).  The planner's batch floors are lowered so that every
    streaming kernel really runs at these batches (BOD_POINTWISE_MIN_M / BOD_STEM_POOL_FUSED_MIN_B).  (Stage 3's 128-channel
    sliding-window kernel re-associates one fp32 addition per output and is therefore off in both runs: its own tests are
    test_sliding_window_3x3_128_channel_kernel_matches_the_generic_launches and tests/test_gpu_conv.py.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, sys, json; sys.path.insert(0, %%r)\n"
            "from bayes_od_rc_amd import synthetic\n"
            "from bayes_od_rc_amd.engine import Engine, make_config\n"
            "eng = Engine(make_config((%d, %d), batch=%d, mc_samples=1))\n"
            "eng.load_weights(synthetic.make_weights())\n"
            "frames = synthetic.make_frames(%d, %d, %d, seed=9)\n" % (hw[0], hw[1], batch, batch, hw[0], hw[1])) + (
            "out = []\n"
            "for rep in range(3):\n"
            "    eng.forward(frames, seed=1, first_image_id=rep)\n"
            "    row = []\n"
            "    for l in range(5):\n"
            "        v = np.ascontiguousarray(eng.get_pyramid(l)).view(np.uint32).ravel()\n"
            "        row.append([int(v.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(v))])\n"
            "    for t in eng.get_raw():\n"
            "        v = np.ascontiguousarray(t).view(np.uint32).ravel()\n"
            "        row.append([int(v.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(v))])\n"
            "    out.append(row)\n"
            "print('CHECKSUMS ' + json.dumps(out))\n")
    code = code % root
    sums = []
    for streaming in ("1", "0"):
        env = dict(os.environ, BOD_POINTWISE=streaming, BOD_SLIDE3X3=streaming, BOD_STEM_POOL_FUSED=streaming, BOD_PW_FUSE_NEXT=streaming,
                   BOD_PW_FUSE_DUAL=streaming, BOD_CHAIN_FUSION="0", BOD_POINTWISE_MIN_M="1", BOD_STEM_POOL_FUSED_MIN_B="1",
                   BOD_SLIDE3X3_C128="0")
        if streaming == "0":
            env["BOD_STEM_SEG64"] = "1"
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("CHECKSUMS ")][0]
        import json
        sums.append(json.loads(line[len("CHECKSUMS "):]))
    assert sums[0] == sums[1], "streaming kernels differ from the generic launches"
    assert sums[0][0][:5] == sums[0][1][:5] == sums[0][2][:5]        # the backbone does not depend on the image id: repeats agree too
