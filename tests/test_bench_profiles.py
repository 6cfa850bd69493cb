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
    tp = bench.wgrad_traffic_per_product()                  # single + grouped weight-gradient launches, per product
    assert isinstance(tp, int) and 150e6 < tp < 800e6
    assert bench.wgrad_traffic_per_product({}) is None
    util = bench.mfma_util_by_kernel()
    assert util and "whole_step" in util and all(0.0 <= v <= 1.0 for v in util.values())
    assert any(k.startswith("gemm_tn_p8_kernel") for k in util)
    ws = bench.wgrad_kernel_split()
    assert ws and ws["kernel_us_per_product"] > 50 and 0.3 < ws["frac_kernel_alone"] < 0.7 and ws["frac_with_reduction"] < ws["frac_kernel_alone"]
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


def test_round6_evidence_set_is_consistent():
    """profiles/r06_final_* come from ONE tools/r06_final.sh call: the per-kernel averages of the --stats summary, of the
    per-step table cut from the SAME trace, and of the un-prefixed summaries bench.py reads must agree (a file regenerated on its own,
    from another box or another build, shows up here)."""
    import csv
    import re
    pdir = os.path.join(ROOT, "profiles")
    stats = {}
    for r in csv.DictReader(open(os.path.join(pdir, "r06_final_seq_kernel_stats.csv"))):
        n = re.sub(r"^void ", "", r["Name"])
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"\(.*$", "", n)[:70]
        stats[n] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    step = {}
    for l in open(os.path.join(pdir, "r06_final_seq_step_kernels.txt")):
        m = re.match(r"^(\S.*?)\s+launches/step\s+([\d.]+)\s+avg\s+([\d.]+) us\s+per-step\s+([\d.]+) ms", l)
        if m:
            step[m.group(1).strip()] = (float(m.group(2)), float(m.group(3)), float(m.group(4)))
    big = [k for k, v in step.items() if v[2] >= 0.5]                 # kernels with >= 0.5 ms per step
    assert len(big) >= 10, big
    for k in big:
        assert k in stats, k
        # the --stats average covers the warm-up steps too: within 3 % of the steady-state window's; 6 % for the kernels with
        # fewer than 12 launches per step (which NT launches take the paired form depends on the step's kept-sample counts, so
        # the populations of gemm_p8_kernel<EPI,256> and gemm_p8_pair_kernel<EPI> differ a little between the two windows)
        tol = 0.03 if step[k][0] >= 12 else 0.06
        assert abs(stats[k][1] / step[k][1] - 1.0) <= tol, (k, stats[k][1], step[k][1])
    # the files bench.py reads are the evidence set's
    for a, b in (("gemm_traffic.json", "r06_final_traffic.json"), ("mfma_util.json", "r06_final_mfma_util.json"),
                 ("raster_traffic.json", "r06_final_raster_traffic.json")):
        assert json.load(open(os.path.join(pdir, a))) == json.load(open(os.path.join(pdir, b))), (a, b)
    # the bench line of the set: its live HIP-event figure for the dominant kernel agrees with the rocprofv3 averages
    bj = json.loads(open(os.path.join(pdir, "r06_final_bench.json")).read().strip().splitlines()[-1])
    # (per weight-gradient PRODUCT: single launches + their reduction passes, and the grouped proj + qkv launches = 2 products)
    tot = lambda k: stats[k][0] * stats[k][1] if k in stats else 0.0
    n_prod = stats["gemm_tn_p8_kernel<true>"][0] + 2 * (stats["gemm_tn_p8_group_kernel"][0] if "gemm_tn_p8_group_kernel" in stats else 0)
    prof = (tot("gemm_tn_p8_kernel<true>") + tot("tn_reduce_kernel") + tot("gemm_tn_p8_group_kernel") + tot("tn_reduce_group_kernel")) / n_prod
    # (round 6, VERDICT item 3: the bench line pools FIVE instrumented steps; its per-product figure and the rocprofv3 averages of the
    # same box must agree within 3 %, and no single instrumented step may sit further than 3 % from the pooled figure)
    assert abs(bj["roofline"]["avg_launch_us"] / prof - 1.0) <= 0.03, (bj["roofline"]["avg_launch_us"], prof)
    assert abs(bj["roofline"]["frac"] - bj["roofline"]["achieved"] / bj["roofline"]["peak"]) < 1e-3
    per_step = bj["roofline"]["frac_per_instrumented_step"]
    # (a step disturbed from outside -- > 10 % below the median of the instrumented steps -- is listed in the line and left out of the
    # pooled figure; at most one in five may be)
    out = {d["step"] for d in bj["roofline"].get("disturbed_steps") or []}
    kept = [f for i, f in enumerate(per_step) if i not in out]
    assert len(kept) >= 3 and len(out) <= len(per_step) // 5 and max(abs(f / bj["roofline"]["frac"] - 1.0) for f in kept) <= 0.03, per_step
    ws = json.load(open(os.path.join(pdir, "wgrad_split.json")))
    assert abs((ws["kernel_us_per_product"] + ws["reduction_us_per_product"]) / prof - 1.0) <= 1e-3
    assert bj["roofline"]["kernel_vs_reduction"] is None or bj["roofline"]["kernel_vs_reduction"]["products"] > 0
    # the line says which library it measured, and the evidence set is the shipped build's
    assert bj["library"]["shipped_build"] is True and bj["library"]["build_flags"] == "" and bj["value"] is not None
    sol = bj["speed_of_light"]
    assert abs(sum(v["floor_ms"] for v in sol["families"].values()) - sol["sum_floor_ms"]) < 0.02
    assert sol["ms_per_step_over_sum_floor"] > 1.0 and sol["perfect_overlap_floor_ms"] <= sol["sum_floor_ms"]
    # config #5 has counters of its own since round 5
    vu = json.load(open(os.path.join(pdir, "r06_final_vitl_mfma_util.json")))
    vt = json.load(open(os.path.join(pdir, "r06_final_vitl_traffic.json")))
    assert any(k.startswith("attn_fwd_win_kernel") for k in vu) and any(k.startswith(("attn_bwd_q_win_kernel", "attn_bwd_qs_win_kernel")) for k in vt)
    assert "raster_bin_keys" in json.load(open(os.path.join(pdir, "r06_final_raster_traffic.json")))
