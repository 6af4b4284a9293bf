"""bench.py's host-side logic that needs no GPU (round 6): the clock / power sampler of the bench line against a fake sysfs tree, and the in-step
roofline arithmetic (time-weighted fraction over the transform kernels of one lane's step) on a synthetic per-kernel list."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _fake_device(tmp_path, pci, sclk_hz=None, power_uw=None, power_name="power1_input", dpm=None):
    d = tmp_path / pci / "hwmon" / "hwmon7"
    d.mkdir(parents=True)
    if sclk_hz is not None:
        (d / "freq1_input").write_text(f"{sclk_hz}\n")
    if power_uw is not None:
        (d / power_name).write_text(f"{power_uw}\n")
    if dpm is not None:
        (tmp_path / pci / "pp_dpm_sclk").write_text(dpm)
    return str(tmp_path)


def test_sampler_reads_hwmon_clock_and_power(tmp_path):
    import bench
    root = _fake_device(tmp_path, "0000:5d:00.0", sclk_hz=2035000000, power_uw=1317000000)
    s = bench.SmiSampler("0000:5d:00.0", period_s=0.005, sysfs=root)
    s.start()
    time.sleep(0.05)
    out = s.stop()
    assert out["samples"] >= 2 and out["sclk_mhz"] == 2035.0 and out["power_w"] == 1317.0 and "note" not in out


def test_sampler_prefers_average_power_and_falls_back_to_dpm_table(tmp_path):
    import bench
    root = _fake_device(tmp_path, "0000:01:00.0", power_uw=900000000, power_name="power1_average", dpm="0: 500Mhz\n1: 1900Mhz *\n2: 2400Mhz\n")
    s = bench.SmiSampler("0000:01:00.0", period_s=10.0, sysfs=root)  # a timed region shorter than one period still reports a reading
    s.start()
    out = s.stop()
    assert out["samples"] == 1 and out["sclk_mhz"] == 1900.0 and out["power_w"] == 900.0


def test_sampler_without_sysfs_files_says_so(tmp_path):
    import bench
    out = bench.SmiSampler("0000:ff:00.0", sysfs=str(tmp_path))
    out.start()
    r = out.stop()
    assert r["samples"] == 0 and "note" in r and "sclk_mhz" not in r
    r = bench.SmiSampler(None).stop()  # no PCI address at all (the placement report failed)
    assert "note" in r


def test_in_step_roofline_is_time_weighted_over_the_transform_kernels():
    import bench
    k = [{"name": "ntt2_kernel<0, 0, 9, 0, 1, 0, 1>", "us": 3000.0, "algorithmic_bytes": 8.0e9, "frac": round(8.0e9 / 3000e-6 / 1e9 / 8000.0, 4), "traffic_ratio": 1.0},
         {"name": "ntt1_inv_kernel<true, false>", "us": 2000.0, "algorithmic_bytes": 6.0e9, "frac": round(6.0e9 / 2000e-6 / 1e9 / 8000.0, 4)},
         {"name": "behz2_extend_kernel<4>", "us": 1000.0, "algorithmic_bytes": 4.0e9, "frac": 0.5}]  # not a transform kernel: in the step's share, not in the fraction
    r = bench.in_step_roofline(k)
    assert r["kernel"] == "ntt2_kernel<0, 0, 9, 0, 1, 0, 1>" and r["share_of_step"] == 0.5 and r["transform_share_of_step"] == round(5000 / 6000, 3)
    assert r["weighted_frac"] == round(14.0e9 / 5000e-6 / 1e9 / 8000.0, 4)
    assert bench.in_step_roofline([k[2]]) is None
