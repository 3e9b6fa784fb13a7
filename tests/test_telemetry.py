"""wf3d.telemetry (board power / clock sampling for bench.py's roofline.power) on a machine without the card: every entry
point degrades to None instead of raising, and the sampler averages what it is given."""
import os

import helpers as H  # noqa: F401  (sys.path)


def test_sampler_without_sensor_is_a_no_op():
    from wf3d import telemetry
    with telemetry.Sampler(None) as s:
        pass
    assert s.watts is None and s.ghz is None
    assert telemetry.power_cap_watts(None) is None


def test_sampler_reads_hwmon_files(tmp_path):
    from wf3d import telemetry
    (tmp_path / "power1_input").write_text("1400000000\n")
    (tmp_path / "freq1_input").write_text("2000000000\n")
    (tmp_path / "power1_cap").write_text("1400000000\n")
    import time
    with telemetry.Sampler(str(tmp_path), period=0.005) as s:
        time.sleep(0.08)
    assert abs(s.watts - 1400.0) < 1e-6 and abs(s.ghz - 2.0) < 1e-9
    assert telemetry.power_cap_watts(str(tmp_path)) == 1400.0
    (tmp_path / "power1_input").write_text("garbage")
    assert telemetry._read(str(tmp_path / "power1_input")) is None
    assert telemetry._read(os.path.join(str(tmp_path), "missing")) is None
