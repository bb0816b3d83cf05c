"""CPU: the bench line's telemetry sampler (bench.Telemetry) on a fake amdgpu hwmon directory -- units, statistics, and that a box
without readable files yields None instead of failing the bench."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _fake_hwmon(tmp_path, power_uw, freq_hz, temp_mc, cap_uw=1400000000):
    d = tmp_path / "hwmon7"
    d.mkdir()
    (d / "power1_input").write_text("%d\n" % power_uw)
    (d / "freq1_input").write_text("%d\n" % freq_hz)
    (d / "temp2_input").write_text("%d\n" % temp_mc)
    (d / "power1_cap").write_text("%d\n" % cap_uw)
    return str(d)


def test_telemetry_reads_power_clock_and_temperature(tmp_path):
    import bench
    d = _fake_hwmon(tmp_path, 1310000000, 2200000000, 53000)
    t = bench.Telemetry(0, hwmon_dir=d)
    t.start()
    time.sleep(0.12)
    with open(os.path.join(d, "power1_input"), "w") as fp:      # the board moves while the region runs
        fp.write("1330000000\n")
    time.sleep(0.12)
    out = t.stop()
    assert out["source"] == d and out["samples"] >= 4 and out["interval_ms"] == 25
    assert 1310.0 <= out["board_power_w_mean"] <= 1330.0 and out["board_power_w_max"] == 1330.0
    assert out["board_power_cap_w"] == 1400.0
    assert out["shader_clock_mhz_mean"] == 2200.0 == out["shader_clock_mhz_min"] == out["shader_clock_mhz_max"]
    assert out["hotspot_temp_c_max"] == 53.0
    assert t.stop() is None                                    # stopped: nothing to report twice


def test_telemetry_is_optional(tmp_path):
    import bench
    empty = tmp_path / "nothing"
    empty.mkdir()
    t = bench.Telemetry(0, hwmon_dir=str(empty))
    t.start()
    time.sleep(0.06)
    assert t.stop() is None                                    # no readable file: no telemetry object, no exception
    none = bench.Telemetry.__new__(bench.Telemetry)
    none.dir, none.samples, none._stop, none._thread = None, [], False, None
    none.start()
    assert none.stop() is None


def test_detection_statistics_aggregates_every_frame():
    """bench.detection_statistics (round 5): the detection-level distance of a precision mode to the CPU leg as a statistic over all
    frames the CPU leg computed -- match rate, order, median / p95 / max of the per-detection errors -- not one frame's anecdote."""
    import numpy as np
    import bench
    rng = np.random.default_rng(0)

    def frame(k, jitter):
        means = np.stack([rng.uniform(50, 400, k), rng.uniform(50, 400, k), rng.uniform(20, 60, k), rng.uniform(20, 60, k)], 1)
        covs = np.tile(np.eye(4) * 4.0, (k, 1, 1))
        scores = rng.random((k, 8)); counts = rng.integers(0, 30, (k, 8)).astype(np.float64)
        ref = (scores, means[:, :, None], covs, counts)
        dev = (scores + jitter * 1e-3, means + jitter, covs * (1 + jitter * 1e-2), counts)
        return dev, ref
    per = []
    for k, j in ((12, 0.0), (9, 0.01), (15, 0.5)):
        dev, ref = frame(k, j)
        p = bench.detection_parity(dev, ref, arrays=True)
        assert p["matched"] == k and p["same_order"]
        per.append(p)
    st = bench.detection_statistics(per + [None])
    assert st["frames"] == 3 and st["matched"] == st["cpu_detections"] == st["device_detections"] == 36 and st["frames_in_same_order"] == 3
    assert st["abs_dmu_px"]["max"] == 0.5 and st["abs_dmu_px"]["median"] <= 0.5 and st["abs_dmu_px"]["p95"] <= 0.5
    assert abs(st["dscore"]["max"] - 5e-4) < 1e-9 and st["rel_dSigma"]["max"] < 6e-3
    assert bench.detection_statistics([]) is None
    e = bench._rel_errs(np.array([1.0, 2.0, 1e-9]), np.array([1.0, 2.002, 0.0]))
    assert abs(e[0] - 0.002 / (2.002 + np.sqrt((1 + 2.002 ** 2) / 3))) < 1e-12 and abs(e[2] - 0.002 / 2.002) < 1e-12      # strict: the zero of `ref` is not rated
