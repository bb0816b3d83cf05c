"""GPU, LAST in the suite (tests/conftest.py orders it): the DESIGN.md 8.4 canaries.

Finding (rounds 5-6): while convolution kernels of this library share a compute unit with post_fuse_kernel / cluster_fuse_kernel, their
4x4 inverses and matrix-vector products came out wrong in lanes 48-63 of a wave -- transiently, uniformly over all XCDs / CUs / SIMDs,
never without such company, never when the two ran on disjoint CUs (tests/tools/selfcheck_probe.py).  Round 6 traced the victim side to
the PACKED fp32 instructions the SLP vectoriser makes of that arithmetic (v_pk_mul_f32 / v_pk_add_f32): csrc/post_kernels.hip and
loss_kernels.hip are built without them (-fno-slp-vectorize), after which 0 of 56 million self-checked waves differ.  No entry point
runs kernels beside each other by default; two handles driven from two host threads do -- these tests ARE that exposure:
  * handle A's posterior + soft-NMS + cluster-and-fuse, re-run on unchanged MC statistics while handle B's forward runs on another
    thread, must reproduce its own arrays bit for bit;
  * handle A's FORWARD (whose convolution epilogues still contain packed fp32 instructions) beside handle B's forward must reproduce
    its pyramid and raw head outputs bit for bit.
A difference is reported as the known issue (xfail: look at 8.4 again), never silently."""
import threading

import numpy as np
import pytest

from conftest import ANCHOR_CFG, BAYES_CFG, NMS_CFG

pytestmark = pytest.mark.gpu


def test_canary_posterior_beside_another_handles_forward_reproduces_itself():
    from bayes_od_rc_amd import synthetic
    from bayes_od_rc_amd.anchor_generator import FpnAnchorGenerator
    from bayes_od_rc_amd.engine import Engine, make_config
    hw, n, batch, iters = (512, 512), 2, 32, 60
    weights = synthetic.make_weights(cls_fg_bias=-1.0)
    anchors = FpnAnchorGenerator(ANCHOR_CFG).generate_all((hw[0], hw[1], 3))
    frames = synthetic.make_frames(batch, hw[0], hw[1], seed=12)
    mk = lambda: Engine(make_config(hw, batch=batch, mc_samples=n, bayes_od_config=BAYES_CFG, nms_config=NMS_CFG, use_full_covar=True))
    a, b = mk(), mk()
    for e in (a, b):
        e.load_weights(weights); e.set_anchors(anchors); e.upload_images(frames)
    a.infer(None, seed=3, first_image_id=0)
    ref = [a.get_posterior(i) for i in range(batch)]
    ref_det = {k: v.copy() for k, v in a.get_detections_batch().items()}
    assert sum(len(r["means"]) for r in ref) > 1000
    b.forward(None); b.synchronize()
    stop = [False]

    def company():
        while not stop[0]:
            b.forward(None, seed=1, first_image_id=0)
            b.synchronize()
    t = threading.Thread(target=company); t.start()
    bad = []
    try:
        for it in range(iters):
            a.posterior(seed=3, first_image_id=0)
            a.nms(); a.cluster_fuse()
            for i in range(batch):
                got = a.get_posterior(i)
                for k in ("means", "covs", "counts", "score", "ranking", "anchor_index"):
                    if not np.array_equal(got[k], ref[i][k]):
                        rows = np.nonzero((np.asarray(got[k]) != np.asarray(ref[i][k])).reshape(len(ref[i][k]), -1).any(axis=1))[0]
                        bad.append((it, i, k, rows[:4].tolist(), len(rows)))
            det = a.get_detections_batch()
            for k in ("scores", "means", "covs", "counts"):
                for i in range(batch):
                    m = ref_det["num"][i]
                    if det["num"][i] != m or not np.array_equal(det[k][i, :m], ref_det[k][i, :m]):
                        bad.append((it, i, "detections." + k, [], 1))
    finally:
        stop[0] = True; t.join()
        a.close(); b.close()
    if bad:
        pytest.xfail("known issue (DESIGN.md 8.4): %d array(s) of the posterior / detections differed beside another handle's forward, e.g. %s"
                     % (len(bad), bad[:3]))


def test_canary_forward_beside_another_handles_forward_reproduces_itself():
    from bayes_od_rc_amd import synthetic
    from bayes_od_rc_amd.engine import Engine, make_config
    hw, n, batch, iters = (512, 512), 2, 16, 25
    weights = synthetic.make_weights()
    mk = lambda seed: (Engine(make_config(hw, batch=batch, mc_samples=n)), synthetic.make_frames(batch, hw[0], hw[1], seed=seed))
    (a, fa), (b, fb) = mk(5), mk(6)
    for e, f in ((a, fa), (b, fb)):
        e.load_weights(weights); e.upload_images(f)
    a.forward(None, seed=3, first_image_id=0)
    ref = [x.copy() for x in a.get_raw()] + [a.get_pyramid(l).copy() for l in range(5)]
    b.forward(None); b.synchronize()
    stop = [False]

    def company():
        while not stop[0]:
            b.forward(None, seed=1, first_image_id=0)
            b.synchronize()
    t = threading.Thread(target=company); t.start()
    bad = []
    try:
        for it in range(iters):
            a.forward(None, seed=3, first_image_id=0)
            got = list(a.get_raw()) + [a.get_pyramid(l) for l in range(5)]
            for k, (g, r) in enumerate(zip(got, ref)):
                if not np.array_equal(g, r):
                    bad.append((it, ("cls", "box", "cov", "p3", "p4", "p5", "p6", "p7")[k], int((g != r).sum())))
    finally:
        stop[0] = True; t.join()
        a.close(); b.close()
    if bad:
        pytest.xfail("known issue (DESIGN.md 8.4): the forward beside another handle's forward differed in %d array(s), e.g. %s" % (len(bad), bad[:4]))
