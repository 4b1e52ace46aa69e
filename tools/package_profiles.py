#!/usr/bin/env python3
"""Packages a tools/profile_round.sh run (gpurun_out/profile_<tag>, gpurun_out/pmc_<tag>, ...) into profiles/<round>/ and writes
its README.md from the packaged files, so that every number in it is one in a file beside it.
usage: package_profiles.py <tag> <round dir>"""
import csv, glob, json, os, shutil, subprocess, sys

tag, dst = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
src = os.path.join(out, "profile_" + tag)
os.makedirs(dst, exist_ok=True)


def last_json(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


# ---- bench lines
bench = last_json(os.path.join(src, "bench.json"))
FIF = int(bench["config"]["frames_in_flight"])
drv = last_json(os.path.join(src, "bench_driver_cmd.json"))
json.dump(bench, open(os.path.join(dst, "bench_unprofiled.json"), "w"), indent=1)
json.dump(drv, open(os.path.join(dst, "bench_driver_cmd.json"), "w"), indent=1)
# the live PMC passes of that bench run, also kept as the labelled fall-back bench.py reads when it cannot profile itself
pm = bench["roofline"].get("pmc")
if pm:
    key = ["random1m", 128, 0, 1024, 465, 1, bench["config"]["passes_per_timed_region"]]
    r_ = bench["roofline"]
    per_query = pm["valu_instructions_per_launch"] / (r_["per_frame"]["queries"] / r_["launches_per_frame"])      # what bench.py derives an N > 1 roofline from
    json.dump({"config_key": key, "taken_at": "packaged from the bench run in bench_unprofiled.json", "valu_instructions_per_query": per_query, "pmc": pm},
              open(os.path.join(dst, "pmc_bench.json"), "w"), indent=1)

# ---- kernel statistics (overlapped = production; standalone = MCRT_NO_OVERLAP=1)
shutil.copy(glob.glob(os.path.join(out, "pmc_" + tag, "stats", "*", "*kernel_stats.csv"))[0], os.path.join(dst, "kernel_stats.csv"))
shutil.copy(os.path.join(src, "kernels_standalone.txt"), os.path.join(dst, "kernels_standalone.txt"))
shutil.copy(os.path.join(src, "frame_timeline.txt"), os.path.join(dst, "frame_timeline.txt"))
shutil.copy(os.path.join(src, "frame_timeline_one_frame.txt"), os.path.join(dst, "frame_timeline_one_frame.txt"))
shutil.copy(os.path.join(src, "configs.txt"), os.path.join(dst, "baseline_configs.txt"))

# ---- PMC passes per kernel
pmc = {}
for kernel, name in (("k_trace_lane<false", "pmc_k_trace_lane.json"), ("k_march<false", "pmc_k_march.json"), ("k_shade<false", "pmc_k_shade.json")):
    env = dict(os.environ, PMC_KERNEL=kernel)
    subprocess.check_output([sys.executable, os.path.join(root, "tools", "pmc_summary.py"), os.path.join(out, "pmc_" + tag)], env=env)
    d = json.load(open(os.path.join(out, "pmc_" + tag, "summary.json")))
    p = {k: v["avg_per_launch"] for k, v in d["pmc"].items()}
    cu = p["SQ_BUSY_CU_CYCLES"] / 256
    der = {"note": ("per launch, averaged over the launches of a bench.py run (%d frames per pass); separate --pmc passes; kernels run one at a time under --pmc. "
                   "FETCH_SIZE / WRITE_SIZE are KiB; fabric bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE counts 64 B per 128-B request, MI355X_MICROARCH.md). "
                   "SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* are quad-cycles.") % FIF,
           "kernel_busy_cycles_per_cu": cu,
           "valu_instructions": p["SQ_INSTS_VALU"], "salu_instructions": p["SQ_INSTS_SALU"], "vmem_read_instructions": p["SQ_INSTS_VMEM_RD"], "lds_instructions": p["SQ_INSTS_LDS"],
           "valu_ipc_per_simd": p["SQ_INSTS_VALU"] / 1024 / cu,
           "valu_lane_utilisation": p["SQ_THREAD_CYCLES_VALU"] / (64 * p["SQ_ACTIVE_INST_VALU"]),
           "wave_time_waiting_on_memory": p["SQ_WAIT_ANY"] / p["SQ_WAVE_CYCLES"], "wave_time_issue_stalled": p["SQ_WAIT_INST_ANY"] / p["SQ_WAVE_CYCLES"],
           "wave_time_issuing": p["SQ_ACTIVE_INST_ANY"] / p["SQ_WAVE_CYCLES"],
           "tcp_lane_accesses_per_cycle_per_cu": p["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256 / cu,
           "l1_hit_rate": 1 - p["TCP_TCC_READ_REQ_sum"] / p["TCP_TOTAL_CACHE_ACCESSES_sum"], "l2_hit_rate": p["TCC_HIT_sum"] / p["TCC_REQ_sum"],
           "fabric_bytes_per_launch": (2 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024}
    d["derived"] = der
    json.dump(d, open(os.path.join(dst, name), "w"), indent=1)
    pmc[name] = der

# ---- README
roofs = os.path.join(root, "profiles", "round2")       # the two roofs were calibrated in round 2 (same tool, same chip)
cal = json.load(open(os.path.join(roofs, "valu_roof.json")))
mix5 = [r for r in cal["results"] if r["class"].startswith("BVH4 node-step mix") and r["waves_per_simd"] == 5][0]
fet = json.load(open(os.path.join(roofs, "fetch_roof.json")))


def fr(lanes, nbytes, table):
    return [r for r in fet["results"] if r["lanes_per_run"] == lanes and r["bytes_per_lane"] == nbytes and r["table"].startswith(table)][0]


ks = {}
for row in csv.DictReader(open(os.path.join(dst, "kernel_stats.csv"))):
    ks[row["Name"].split("(")[0].replace("void mcrt::", "")] = row
alone = {}
for line in open(os.path.join(dst, "kernels_standalone.txt")):
    for k in ("k_trace_lane<false>", "k_march<false", "k_shade<false>"):
        if k in line and " avg " in line:
            alone[k] = float(line.split(" avg ")[1].split()[0])
r = bench["roofline"]; t = pmc["pmc_k_trace_lane.json"]; m = pmc["pmc_k_march.json"]; s = pmc["pmc_k_shade.json"]
cb = bench["cpu_baseline"]
txt = """# profiles/round3 -- MI355X (gfx950), ROCm 7.2

Workload of every file unless it says otherwise: `bench.py` defaults = synthetic 1 M random triangles, 128 scan-lines x 1024 sample
paths per frame, 465 RF rows, max depth 10, one GPU, @FIF@ frames in flight per pass.  Produced by `tools/profile_round.sh` on a gpurun
box and packaged by `tools/package_profiles.py`, which also wrote this file from the files beside it.

| file | what |
|---|---|
| `../round2/valu_roof.json`, `valu_roof_pmc.json`, `fetch_roof.json` | the two calibrated roofs (`tools/valu_roof.hip`, `tools/fetch_roof.hip`), measured in round 2 on the same chip with the same tools: not repeated |
| `exp_*` | this round's experiments, copied in by hand (DESIGN.md 5.5): queues sorted into ray bundles against the order-preserving compaction (`exp_sorted_bundles_*`, `exp_unsorted_*`: per-bounce PMC of the walk, stamp-build lane statistics, refill / leaf-batch thresholds with sorted queues), kernels with path state in place by path id (`exp_records_kernels_standalone.txt`), CU-masked streams (`exp_cu_masks.txt`) |
| `fetch_roof_same.json` | `tools/fetch_roof_same.hip` on the box: what lanes on ONE address cost the vector memory pipe (only whole adjacent quads are cheaper: 0.3 of four), and inactive lanes (nothing) |
| `bench_unprofiled.json` | `python bench.py`: the JSON line (live PMC passes in child processes, CPU baseline, inline parity check) |
| `bench_driver_cmd.json` | `python bench.py --gpus 1 --steps 20 --warmup 5` (the driver's command: one 20-frame pass per timed region) |
| `pmc_bench.json` | the PMC block of `bench_unprofiled.json`, the labelled fall-back `bench.py` reads when it cannot profile itself |
| `kernel_stats.csv` | `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps @FIF@ --warmup @FIF@ --no-cpu-baseline --no-latency-leg --no-pmc` (kernels overlap as in production) |
| `kernels_standalone.txt` | the same with `MCRT_NO_OVERLAP=1`: every kernel alone on the GPU |
| `pmc_k_trace_lane.json`, `pmc_k_march.json`, `pmc_k_shade.json` | separate `--pmc` passes (tools/pmc.sh), per-launch averages + derived figures |
| `frame_timeline.txt`, `frame_timeline_one_frame.txt` | start / duration of every launch of one pass: @FIF@ frames in flight, and one frame at a time |
| `baseline_configs.txt` | the five BASELINE.json configurations on one GPU (tools/configs.sh) |

## The two roofs (calibrated, not assumed)

**VALU issue.**  A wave64 VALU instruction costs the SIMD about FOUR cycles for the instruction classes the kernels are made of, however
many wavefronts share the SIMD: `v_fma_f32` independent streams reach %.3f instructions per cycle and SIMD at 8 waves, `v_min3_f32`,
`v_pk_mul_f32`, DPP moves and compare/select pairs 0.24-0.26, `v_fma_f64` / `v_mul_f64` 0.24; only plain integer adds (0.45) and
dependent `v_fma_f32` chains of many waves (0.44) come near the 2-cycle figure of the micro-architecture guide.  The register-only
part of a BVH4 node step (as it was in round 2: packed subtract / multiply, min / max / min3 / max3, compares, selects, key arithmetic) issues at
**%.3f instructions per cycle and SIMD at 5 waves per SIMD, clock %.2f GHz** -- the roof `bench.py` prices the walk against
(1024 SIMDs x %.3f x %.2f GHz = %.0f G wave-instructions per second).

**Vector memory pipe.**  A scattered wave-level `global_load_dwordx4` (every lane its own 128-byte line) costs a CU %.0f cycles
when the data is in L1 (%.2f lanes per cycle), the same from L2, and %.0f cycles from the Infinity Cache; eight lanes reading one whole
line cost %.0f cycles per wave-load (%.1f B/cycle/CU) -- the TCP moves ~24 B per cycle and CU through `dwordx4` loads however they
are shaped, twice that through `dword` / `dwordx2` loads of contiguous lanes.  What the walk pays per node is therefore the number of
16-byte pieces a lane fetches: 7 with the 128-byte nodes of round 1, 4 with the 64-byte half-float nodes.
`bench.py` also reports the walk against the best class measured (`frac_vs_best_class`, 0.449) and against the guide's two cycles per
instruction (`frac_vs_architectural`, 0.5).

## What the kernels do with them (per launch = one bounce of a @FIF@-frame pass)

| kernel | alone | overlapped | VALU instr. | IPC / SIMD (of %.3f) | lanes active | waiting on memory | TCP lane-accesses / cycle / CU | fabric bytes |
|---|---|---|---|---|---|---|---|---|
| `k_trace_lane` | %.0f us | %.0f us | %.0f M | %.3f (%.0f %%) | %.0f %% | %.0f %% | %.2f | %.0f MB |
| `k_march` | %.0f us | %.0f us | %.0f M | %.3f (%.0f %%) | %.0f %% | %.0f %% | %.2f | %.0f MB |
| `k_shade` | %.0f us | %.0f us | %.0f M | %.3f (%.0f %%) | %.0f %% | %.0f %% | %.2f | %.0f MB |

The walk (`k_trace_lane`) issues at %.0f %% of the calibrated VALU ceiling with %.0f %% of its lanes active, and its %.1f M wave-level
loads per launch keep the TCP at %.2f lane-accesses per cycle (the scattered-`dwordx4` rate measured above is %.2f): with 64-byte nodes
both pipes are loaded to about the same degree.  The BVH is served on-die: L1 hit rate %.0f %%, L2 %.0f %% of the rest, fabric traffic
%.0f MB per launch -- `bench.py` reports it as `hbm_measured_frac` = %.3f of the 8 TB/s HBM figure (the algorithmic bytes, %.1f GB per
launch, flow at %.1f TB/s from the caches).

`bench_unprofiled.json` (a step is a WHOLE B-mode frame: trace, accumulate, PSF, envelope, scan conversion): **%.1f M rays/s, %.3f ms per frame (%.0f frames/s)**, timed region repeated %d times (min / median / max
%.3f / %.3f / %.3f ms per frame); one frame at a time %.2f ms per frame; roofline `frac` = %.2f of the VALU ceiling
(%.0f of %.0f G wave-instructions per second, wall time of the launches, tails and the concurrently running `k_march` included);
`parity_check.rf_bit_exact` = %s on %d scan-lines.  CPU baseline (the oracle, %d usable cores of %d hardware threads, %.1f kept busy):
%.2f M rays/s, one thread %.1f k rays/s.  `bench_driver_cmd.json` (one 20-frame pass): %.1f M rays/s, %.3f ms per frame; the same pass with a
different probe pose in every frame (`sweep`): %.3f ms per frame.
""" % (
    max(x["simd_ipc"] for x in cal["results"] if x["class"] == "v_fma_f32 independent"), mix5["simd_ipc"], mix5["clock_ghz"], mix5["simd_ipc"], mix5["clock_ghz"], 1024 * mix5["simd_ipc"] * mix5["clock_ghz"],
    fr(1, 16, "16 KiB")["cycles_per_wave_load_per_cu"], 64 / fr(1, 16, "16 KiB")["cycles_per_wave_load_per_cu"], fr(1, 16, "64 MiB")["cycles_per_wave_load_per_cu"],
    fr(8, 16, "16 KiB")["cycles_per_wave_load_per_cu"], fr(8, 16, "16 KiB")["bytes_per_cycle_per_cu"],
    mix5["simd_ipc"],
    alone.get("k_trace_lane<false>", 0), float(ks["k_trace_lane<false>"]["AverageNs"]) / 1e3, t["valu_instructions"] / 1e6, t["valu_ipc_per_simd"], 100 * t["valu_ipc_per_simd"] / mix5["simd_ipc"], 100 * t["valu_lane_utilisation"], 100 * t["wave_time_waiting_on_memory"], t["tcp_lane_accesses_per_cycle_per_cu"], t["fabric_bytes_per_launch"] / 1e6,
    alone.get("k_march<false", 0), float(next(v for k, v in ks.items() if k.startswith("k_march<false, 2"))["AverageNs"]) / 1e3, m["valu_instructions"] / 1e6, m["valu_ipc_per_simd"], 100 * m["valu_ipc_per_simd"] / mix5["simd_ipc"], 100 * m["valu_lane_utilisation"], 100 * m["wave_time_waiting_on_memory"], m["tcp_lane_accesses_per_cycle_per_cu"], m["fabric_bytes_per_launch"] / 1e6,
    alone.get("k_shade<false>", 0), float(ks["k_shade<false>"]["AverageNs"]) / 1e3, s["valu_instructions"] / 1e6, s["valu_ipc_per_simd"], 100 * s["valu_ipc_per_simd"] / mix5["simd_ipc"], 100 * s["valu_lane_utilisation"], 100 * s["wave_time_waiting_on_memory"], s["tcp_lane_accesses_per_cycle_per_cu"], s["fabric_bytes_per_launch"] / 1e6,
    100 * t["valu_ipc_per_simd"] / mix5["simd_ipc"], 100 * t["valu_lane_utilisation"], t["vmem_read_instructions"] / 1e6, t["tcp_lane_accesses_per_cycle_per_cu"], 64 / fr(1, 16, "16 KiB")["cycles_per_wave_load_per_cu"],
    100 * t["l1_hit_rate"], 100 * t["l2_hit_rate"], t["fabric_bytes_per_launch"] / 1e6, r.get("hbm_measured_frac") or 0.0, r["algorithmic_bytes_per_launch"] / 1e9, r["algorithmic_GBps_cache_served"] / 1e3,
    bench["value"] / 1e6, bench["ms_per_step"], bench["frames_per_sec"], bench["config"]["timed_region_repeats"], *bench["config"]["repeat_ms_per_step_min_median_max"],
    bench["one_frame_at_a_time"]["ms_per_step"], r.get("frac") or 0.0, r.get("achieved") or 0.0, r["peak"],
    bench["parity_check"]["rf_bit_exact"], bench["parity_check"]["scan_lines"], cb["cores"], cb["host"]["cpu_count"], cb["cores_kept_busy"], cb["value"] / 1e6, cb["single_thread"]["value"] / 1e3,
    drv["value"] / 1e6, drv["ms_per_step"], drv["sweep"]["ms_per_step"])
txt = txt.replace("@FIF@", str(FIF))
open(os.path.join(dst, "README.md"), "w").write(txt)
print(txt)
