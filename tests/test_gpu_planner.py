"""GPU: the forward plan's kernel choices by batch and geometry (feature_extractor.py:195-213: the same block at every batch).
The streaming kernels of the backbone -- sliding-window 3x3, pointwise 1x1, fused stem + pool -- are chosen by WORKGROUPS against
compute units (round 3 chose by pixel count: 6 workgroups on 256 CUs at 3 frames).  For the batches on either side of the planner's thresholds (512x512: 1, 3, 256; 384x1248: 128; the hand-run sweep: every
batch in {1, 3, 8, 32, 128, 256} x both geometries) the default plan must never be more than 3 % slower than the plan with one of them switched off -- nor,
since round 4, than the plan with one of that round's changes undone (tests/tools/planner_sweep.py lists the switches)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))


# (round 6, suite time: the test sweeps the batches on either side of every planner threshold -- sliding window from 1.5 workgroups per
#  CU, pointwise from 128 output pixels per CU, fused stem + pool from one image per CU -- per geometry; tests/tools/planner_sweep.py
#  run by hand covers B in {1, 3, 8, 32, 128, 256} x both: 96 child processes, 145 s of the suite's clock)
@pytest.mark.parametrize("hw,batches", [((512, 512), (1, 3, 256)), ((384, 1248), (128,))])
def test_default_plan_is_never_beaten_by_an_alternative(hw, batches):
    import planner_sweep
    rows = planner_sweep.sweep(batches=batches, geoms=(hw,))
    assert [r["batch"] for r in rows] == list(batches)
    bad = []
    for r in rows:
        for alt in [k for k in r if k not in ("hw", "batch", "default")]:
            # 3 % of the forward, with a floor of 30 us for the sub-millisecond forwards (run-to-run noise of a 1 ms measurement)
            if r["default"] > r[alt] * 1.03 + 0.03:
                bad.append((r["batch"], alt, r["default"], r[alt]))
    assert not bad, bad


def test_small_batches_do_not_take_the_sliding_window_kernel(monkeypatch):
    """3 frames of 512x512 = 6 column strips: the generic kernel (the sweep above measures the consequence; this pins the rule itself
    through the per-op trace: no launch of that batch may take > 100 us in stage 2's 3x3 layers)."""
    import subprocess
    code = r'''
import sys; sys.path.insert(0, %r)
from bayes_od_rc_amd import synthetic
from bayes_od_rc_amd.engine import Engine, make_config
eng = Engine(make_config((512, 512), batch=3, mc_samples=1))
eng.load_weights(synthetic.make_weights())
eng.upload_images(synthetic.make_frames(3, 512, 512, seed=1))
for _ in range(4): eng.forward(None)
eng.synchronize()
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env["BOD_TRACE_OPS"] = "4"
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [l.split() for l in out.stderr.splitlines() if l.startswith("res2") and "_branch2b" in l.split()[0]]
    assert len(rows) == 3, out.stderr[-3000:]
    for r in rows:
        assert float(r[-2]) < 0.1, r            # ms of the launch (172 us on the sliding-window kernel with 6 workgroups)
