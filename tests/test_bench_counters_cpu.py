"""bench.py's roofline object quotes committed counter files (PMC passes cannot run inside a bench run): the files it looks for exist for
the three key-switch configurations, carry the fields the line uses, and are consistent with themselves."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_counter_files_feed_the_bench_line():
    b = _bench()
    for config, ops in (("mul_relin_rescale", 1024), ("dot", 64), ("bfv_matmul", 64)):
        got = b.committed_counters(config, 1, ops)
        assert got, config
        j, path = got
        assert path.startswith("profiles/r") and path.endswith(f"_{config}_kernel_bounds.json")
        assert j["ops_per_step"] == ops
        assert j["hbm_bytes_per_op"] > 0 and j["valu_wave_instr_per_op"] > 0 and 1000 < j["sustained_mhz_time_weighted"] < 2500
        ks = j["kernels"]
        assert abs(sum(k["valu_wave_instr"] for k in ks) - j["valu_wave_instr_per_step"]) <= 1e-6 * j["valu_wave_instr_per_step"]
        for k in ks:  # a fraction above ~1 would mean a broken unit somewhere (GRBM under-counts on sub-millisecond dispatches: allow 1.1)
            assert 0 <= k["hbm_frac_of_8TBps"] < 1.0 and 0 <= k["valu_issue_frac"] < 1.1, (config, k["kernel"])
    assert b.committed_counters("eltwise_mul", 1, 256) is None  # streaming configurations: SURVEY 8d's HBM roofline, no counter file
    # the headline's algorithmic bytes and the traffic ratio the line prints
    j, _ = b.committed_counters("mul_relin_rescale", 1, 1024)
    assert 5.5 < j["hbm_bytes_per_op"] / 24780800 < 7.5


def test_power_sample_reads_the_devices_hwmon_or_reports_nothing(tmp_path, monkeypatch):
    """bench.py's `roofline.power`: without the device's hwmon files the line simply carries no power block (never an exception); with them
    (a stand-in directory tree here) it reports median / maximum watts of the run's last two thirds against the cap."""
    import glob as globmod
    import types
    b = _bench()
    props = types.SimpleNamespace(pci_domain_id=0, pci_bus_id=0xfe, pci_device_id=0x1f)
    ran = []
    assert b.sample_power(props, lambda k: ran.append(k), 0.01, window_seconds=0.05) is None and not ran  # no such PCI function here
    hw = tmp_path / "hwmon" / "hwmon3"
    hw.mkdir(parents=True)
    (hw / "power1_input").write_text("1370000000\n")
    (hw / "power1_cap").write_text("1400000000\n")
    (hw / "freq1_input").write_text("1975000000\n")
    real_glob = globmod.glob
    monkeypatch.setattr(globmod, "glob", lambda pat: [str(hw)] if pat.endswith("/hwmon/hwmon*") else real_glob(pat))
    import time
    got = b.sample_power(props, lambda k: (ran.append(k), time.sleep(0.4)), 0.1, window_seconds=0.4)
    assert ran == [4] and got["package_w_median"] == 1370.0 and got["cap_w"] == 1400.0 and got["frac_of_cap"] == 0.979
    assert got["sclk_mhz_median"] == 1975.0 and got["samples"] >= 3
