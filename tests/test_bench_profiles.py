"""-m "not gpu": every key bench.py reads from the committed PMC summaries under profiles/ exists there.
(Round 2 lost the rasterizer's `traffic` in the driver's record because a profile's layout changed under the reader.)"""
import json
import os

from conftest import ROOT


def test_profile_readers_find_their_keys():
    import bench
    per_sample = bench.raster_traffic_per_sample()
    algorithmic = 32 * 1_000_000 + 3 * 480 * 640
    assert per_sample is not None and algorithmic <= per_sample <= 2.5 * algorithmic       # traffic >= algorithmic bytes
    t = bench.gemm_traffic_per_launch("gemm_tn_p8_kernel")
    assert isinstance(t, int) and t > 100e6
    util = bench.mfma_util_by_kernel()
    assert util and "whole_step" in util and all(0.0 <= v <= 1.0 for v in util.values())
    assert any(k.startswith("gemm_tn_p8_kernel") for k in util)
    clk = bench.kernel_clock()
    assert clk and 1.0 < clk["gemm_p8_ghz"] <= 2.4 and 1.0 < clk["gemm_tn_p8_ghz"] <= 2.4
    assert abs(clk["at_clock_peak_tflops"] - 2500.0 * min(clk["gemm_p8_ghz"], clk["gemm_tn_p8_ghz"]) / 2.4) < 1.0


def test_profile_readers_report_a_missing_layout_as_none():
    import bench
    assert bench.raster_traffic_per_sample({"hbm_bytes_per_sample": 1}) is None            # the pre-r02 layout
    assert bench.raster_traffic_per_sample({"raster_bin_keys": {"hbm_bytes_per_launch": 10},
                                            "raster_bin_accum": {"hbm_bytes_per_launch": 6},
                                            "_meta": {"samples_per_launch": 4}}) == 4.0
    assert bench.gemm_traffic_per_launch("nope", {}) is None
    assert bench.kernel_clock({"kernels": {}}) is None


def test_every_profile_json_parses():
    pdir = os.path.join(ROOT, "profiles")
    names = [n for n in os.listdir(pdir) if n.endswith(".json")]
    assert names
    for n in names:
        json.load(open(os.path.join(pdir, n)))
