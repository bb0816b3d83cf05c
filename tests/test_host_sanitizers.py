"""Host sanitizer build (race detection / sanitizers): the host-side index tables of the head towers
(bayes-od-rc_amd/csrc/plan_tables.h -- the code bod_create runs) compiled with -fsanitize=address,undefined, and a CPU replay of
the row-reuse kernel's index arithmetic against them (tests/host/plan_tables_check.cpp).  GPU AddressSanitizer is not available on
this pool; the device side is covered by the parity tests."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plan_tables_under_asan_ubsan(tmp_path):
    gxx = shutil.which("g++")
    assert gxx, "g++ is part of the image"
    exe = str(tmp_path / "plan_tables_check")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall", "-Wextra", "-Werror",
           "-I" + os.path.join(ROOT, "bayes-od-rc_amd", "csrc"), os.path.join(ROOT, "tests", "host", "plan_tables_check.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failures" in r.stdout and "configurations" in r.stdout, r.stdout
    assert int(r.stdout.split("plan_tables_check:")[1].split()[0]) >= 50


def test_engine_uses_the_checked_tables():
    """engine.hip must build its tables with the functions the sanitizer run checks, not with a private copy."""
    src = open(os.path.join(ROOT, "bayes-od-rc_amd", "csrc", "engine.hip")).read()
    for fn in ("head_row_tables(", "xr_tile_rows(", "xr_tile_rows_aggregated(", "pyramid_geometry("):
        assert fn in src, fn
    assert "struct RowEnt {" not in open(os.path.join(ROOT, "bayes-od-rc_amd", "csrc", "kernels.h")).read()
