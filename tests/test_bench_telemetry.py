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
