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
    with open(os.path.join(ROOT, "profiles", "round4", "pmc_bench.json")) as f:
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
    r = b.roofline_from(d, 0.75, 6000.0)
    assert r["bound"] == "valu" and r["derived"] is True and r["traffic"] is None and r["hbm_measured_frac"] is None
    assert 0.0 < r["frac_vs_architectural"] < r["frac_vs_best_class"] < r["frac"] < 1.5
    live = dict(committed["pmc"])
    r2 = b.roofline_from(live, 3.4, 9000.0)
    assert r2["derived"] is False and r2["traffic"] == live["traffic_bytes_per_launch"] and 0.0 < r2["hbm_measured_frac"] < 0.2
    if "tcp_lane_accesses_per_launch" in live:
        # the second roof is a TIME bound (counted accesses x the cheapest measured cost per access): it cannot exceed the launch's duration
        assert 0.3 < r2["second_roof"]["frac"] <= 1.02 and r2["binding_roof"] in ("vector memory pipe (second_roof)", "valu issue (frac)")
        assert list(r2)[0] == "frac_vs_architectural" and r2["peak_source"]["source"].startswith("profiles/round4")
    empty = b.roofline_from(None, 1.0, 1.0)
    assert empty["frac"] is None and empty["achieved"] is None and empty["traffic"] is None


def test_usable_cores_is_bounded_by_the_machine():
    b = _bench()
    n, info = b.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1) and info["cpu_count"] >= n
