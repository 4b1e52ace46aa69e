"""bench.py's host-side helpers (no GPU): the pass split of a timed region, the roofline block, and the committed counter fall-back
that gives an N > 1 line its (derived) roofline."""
import argparse
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_passes_are_the_fewest_and_even():
    b = _bench()
    assert b.passes(20, 128) == [20] and b.passes(128, 128) == [128] and b.passes(0, 16) == []
    assert b.passes(20, 16) == [10, 10] and b.passes(300, 128) == [100, 100, 100] and b.passes(7, 3) == [3, 2, 2]
    for count in range(1, 70):
        for F in (1, 3, 16, 64):
            p = b.passes(count, F)
            assert sum(p) == count and max(p) <= F and max(p) - min(p) <= 1 and len(p) == -(-count // F)


def test_roofline_block_and_the_derived_fallback():
    b = _bench()
    with open(os.path.join(ROOT, "profiles", b.PMC_ROUND, "pmc_bench.json")) as f:
        committed = json.load(f)
    args = argparse.Namespace(workload="random1m", scanlines=128, scanlines_total=0, rays=1024, rows=465, gpus=1)
    same = b.committed_pmc(args, committed["config_key"][6], 1.0)
    assert same["valu_instructions_per_launch"] == committed["pmc"]["valu_instructions_per_launch"] and not same.get("derived")
    # another pass size / rank count: instructions per closest-hit query x the queries counted per launch, labelled as derived
    args2 = argparse.Namespace(workload="random1m", scanlines=128, scanlines_total=0, rays=1024, rows=465, gpus=8)
    d = b.committed_pmc(args2, [20], 1.0e6)
    assert d["derived"] is True and d["traffic_bytes_per_launch"] is None
    assert abs(d["valu_instructions_per_launch"] - committed["valu_instructions_per_query"] * 1.0e6) < 1.0
    assert 100.0 < committed["valu_instructions_per_query"] < 5000.0
    assert b.committed_pmc(argparse.Namespace(workload="sphere", scanlines=16, scanlines_total=0, rays=64, rows=465, gpus=2), [4], 10.0) is None
    r = b.roofline_from(d, 0.75, 6000.0, 4.5e9)
    assert r["bound"] == "valu" and r["derived"] is True and r["traffic"] is None and r["hbm_measured_frac"] is None
    # the block LEADS with the guide-anchored figure (2 cycles per wave64 instruction); the self-calibrated roof of rounds 2-4 is secondary
    assert r["frac"] == r["frac_vs_architectural"] and 0.0 < r["frac"] < r["calibrated"]["frac_vs_best_class"] < r["calibrated"]["frac"] < 1.5
    assert abs(r["peak"] - 1024 * 0.5 * r["calibrated"]["peak_source"]["clock_ghz"]) < 1e-6 and r["achieved"] / r["peak"] == r["frac"]
    assert r["hbm"]["algorithmic_over_hbm_peak"] == 6000.0 / 8000.0 and r["hbm"]["algorithmic_bytes_per_launch"] == 4.5e9
    live = dict(committed["pmc"])
    r2 = b.roofline_from(live, 3.4, 9000.0, 4.5e9)
    assert r2["derived"] is False and r2["traffic"] == live["traffic_bytes_per_launch"] and 0.0 < r2["hbm_measured_frac"] < 0.2
    assert r2["hbm"]["measured_frac"] == r2["hbm_measured_frac"] and 0.0 < r2["hbm"]["traffic_over_algorithmic"] < 1.0
    assert 0.0 < r2["frac_lane_level"] < r2["frac"]
    if "tcp_lane_accesses_per_launch" in live:
        # the second roof is a TIME bound (counted accesses x the cheapest measured cost per access): it cannot exceed the launch's duration
        assert 0.3 < r2["second_roof"]["frac"] <= 1.02 and r2["binding_roof"] in ("vector memory pipe (second_roof)", "valu issue (calibrated mix)")
        assert isinstance(r2["second_roof"]["consistent_pair"], bool) and r2["second_roof"]["pair_note"]
        assert r2["calibrated"]["peak_source"]["source"].startswith("profiles/round4")
    empty = b.roofline_from(None, 1.0, 1.0, 1.0)
    assert empty["frac"] is None and empty["achieved"] is None and empty["traffic"] is None
    # the other kernels of a bounce in the walk's terms
    kb = b.kernel_block("k_march", "x", 1.2, 0.5, {"valu_instructions_per_launch": 2.0e8, "lane_utilisation": 0.8, "traffic_bytes_per_launch": 1.0e8, "tcp_lane_accesses_per_launch": 1e8}, 3.0e8, 2.4)
    assert abs(kb["dilation_beside_the_rest"] - 2.4) < 1e-9 and 0 < kb["valu_frac_vs_architectural_overlapped"] < kb["valu_frac_vs_architectural_alone"] < 1
    assert abs(kb["valu_frac_vs_architectural_alone"] - 2.0e8 / 0.5e-3 / 1e9 / (1024 * 0.5 * 2.4)) < 1e-12


def test_usable_cores_is_bounded_by_the_machine():
    b = _bench()
    n, info = b.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1) and info["cpu_count"] >= n
