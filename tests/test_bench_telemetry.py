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


def test_detections_outside_1e3_are_counted_by_cause():
    """bench.explain_detection / detection_statistics (round 6): a detection that differs from the CPU leg's by more than 1e-3 is reported
    with WHY -- another centre, an anchor kept by one side's filter only, a cluster member across the affinity threshold, a member's
    sampled counts, or arithmetic -- and the gate's maxima are taken over the detections whose discrete decisions agree."""
    import numpy as np
    import bench

    def ctx_pair():
        # five kept anchors: centre 0 with members {0, 1}, centre 3 with members {3, 4}; anchor 2 is nobody's member
        means = np.array([[100, 100, 40, 40], [102, 101, 40, 40], [300, 300, 40, 40], [200, 200, 40, 40], [201, 202, 40, 40]], np.float32)
        anchor = np.array([10, 11, 12, 13, 14])
        counts = np.tile(np.arange(8, dtype=np.float32) + 0.125, (5, 1))
        corners = bench._vuhw_corners32(means)
        iou = np.stack([bench._iou_column_a15(corners, c) for c in range(5)], 1)
        cpu = {"anchor_index": anchor.copy(), "centres": np.array([0, 3]), "iou": iou, "counts": counts.copy()}
        dev = {"anchor_index": anchor.copy(), "centres": np.array([0, 3]), "means": means.copy(), "counts": counts.copy()}
        return dev, cpu
    dev, cpu = ctx_pair()
    assert bench.explain_detection(0, 0, dev, cpu) == "numeric" and bench.explain_detection(1, 1, dev, cpu) == "numeric"
    dev, cpu = ctx_pair(); dev["counts"][1, 0] += 1; dev["counts"][1, 1] -= 1
    assert bench.explain_detection(0, 0, dev, cpu) == "draw_flip" and bench.explain_detection(1, 1, dev, cpu) == "numeric"
    dev, cpu = ctx_pair(); dev["means"][4] = [260, 260, 40, 40]                # member 14 leaves centre 13's cluster on the device
    assert bench.explain_detection(1, 1, dev, cpu) == "member_flip"
    dev, cpu = ctx_pair()                                                      # the device's filter dropped anchor 11 (a member of centre 10)
    for k in ("anchor_index", "means", "counts"):
        dev[k] = np.delete(dev[k], 1, axis=0)
    dev["centres"] = np.array([0, 2])
    assert bench.explain_detection(0, 0, dev, cpu) == "filter_flip" and bench.explain_detection(1, 1, dev, cpu) == "numeric"
    dev, cpu = ctx_pair(); dev["centres"] = np.array([1, 3])                   # soft-NMS chose anchor 11 as the first cluster's centre
    assert bench.explain_detection(0, 0, dev, cpu) == "centre_differs"
    assert bench.explain_detection(0, None, dev, cpu) == "centre_differs"      # an unmatched CPU detection whose centre the device kept
    assert bench.explain_detection(0, 0, None, cpu) is None

    # the statistic: one frame, two detections, the second one moved by a member flip
    dev, cpu = ctx_pair(); dev["means"][4] = [260, 260, 40, 40]
    k = 2
    ref = (np.full((k, 8), 0.125), np.array([[101, 100.5, 40, 40], [200.5, 201, 40, 40]], np.float64)[:, :, None], np.tile(np.eye(4) * 4.0, (k, 1, 1)), np.ones((k, 8)))
    det = (ref[0].copy(), ref[1][:, :, 0].copy(), ref[2].copy(), ref[3].copy())
    det[1][1] += 0.5                                                            # 0.5 px on a 200 px coordinate: 2.5e-3
    p = bench.detection_parity(det, ref, arrays=True, dev_ctx=dev, cpu_ctx=cpu)
    assert p["_cause"] == ["numeric", "member_flip"] or p["_cause"] == ["member_flip", "numeric"]
    st = bench.detection_statistics([p])
    assert st["outside_1e-3"] == {"detections": 1, "of": 2, "by_cause": {"member_flip": 1}}
    assert st["discrete_flips"]["detections"] == 1 and st["discrete_flips"]["by_kind"]["member_flip"] == 1
    assert st["numeric_only"]["detections"] == 1 and st["numeric_only"]["max_rel_dmu"] == 0.0 and st["numeric_only"]["outside_1e-3"] == 0
    assert st["numeric_only"]["max_rms_dSigma"] == 0.0 and st["numeric_only"]["covariance_entries_1pct_floor"] == {"max": 0.0, "outside_1e-3": 0}
    assert st["rel_dmu"]["max"] > 1e-3 and st["fro_dSigma"]["max"] == 0.0
