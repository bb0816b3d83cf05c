"""Tracing hook: with BOD_ROCTX=1 the engine brackets every stage of a step with a roctx range that
rocprofv3 --marker-trace records; without the variable no roctx call is made."""
import csv
import glob
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

STAGES = ["bod:stem", "bod:res2", "bod:res3", "bod:res4", "bod:res5", "bod:fpn", "bod:head_tower_layer_0", "bod:head_tower_layer_1",
          "bod:head_tower_layer_2", "bod:head_tower_layer_3", "bod:posterior", "bod:nms", "bod:cluster_fuse", "bod:infer_async", "bod:collect"]


def test_stage_ranges_reach_rocprofv3(tmp_path):
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    assert os.path.exists(rocprof), "rocprofv3 is part of the ROCm image"
    out = str(tmp_path / "mk")
    env = dict(os.environ, BOD_ROCTX="1", TMPDIR=str(tmp_path))
    # (the program itself after `--`: the profiler's preloaded library initialises the GPU before the program starts)
    cmd = [rocprof, "--marker-trace", "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "m", "--",
           sys.executable, os.path.join(ROOT, "tests", "tools", "marker_demo.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=600)
    assert r.returncode == 0 and "marker_demo done" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    files = glob.glob(os.path.join(out, "**", "*marker_api_trace.csv"), recursive=True)
    assert files, os.listdir(out)
    rows = list(csv.DictReader(open(files[0])))
    names = [row["Function"] for row in rows]
    for st in STAGES:
        assert names.count(st) == 3, (st, names.count(st))            # marker_demo.py runs three steps
    # the forward stages of a step nest inside its bod:infer_async range, in network order
    step = [row for row in rows if row["Function"].startswith("bod:")]
    first = [row["Function"] for row in sorted(step, key=lambda q: int(q["Start_Timestamp"]))][:12]
    assert first[0] == "bod:infer_async" and first[1:11] == STAGES[:10], first
    outer = next(row for row in step if row["Function"] == "bod:infer_async")
    inner = [row for row in step if row["Function"] == "bod:head_tower_layer_1"][0]
    assert int(outer["Start_Timestamp"]) <= int(inner["Start_Timestamp"]) and int(inner["End_Timestamp"]) <= int(outer["End_Timestamp"])


def test_no_marker_library_without_the_switch():
    code = ("import os, sys; sys.path.insert(0, %r); os.environ.pop('BOD_ROCTX', None)\n"
            "from bayes_od_rc_amd import synthetic\n"
            "from bayes_od_rc_amd.engine import Engine, make_config\n"
            "eng = Engine(make_config((96, 96), batch=1, mc_samples=2)); eng.load_weights(synthetic.make_weights())\n"
            "eng.upload_images(synthetic.make_frames(1, 96, 96)); eng.forward(None); eng.synchronize()\n"
            "print('roctx' in open('/proc/self/maps').read())\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip().splitlines()[-1] == "False", r.stdout
