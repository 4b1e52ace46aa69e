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
shutil.copy(sorted(glob.glob(os.path.join(out, "pmc_" + tag, "stats", "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1], os.path.join(dst, "kernel_stats.csv"))   # (the newest run: gpurun merges into the local directory)
shutil.copy(os.path.join(src, "kernels_standalone.txt"), os.path.join(dst, "kernels_standalone.txt"))
shutil.copy(os.path.join(src, "frame_timeline.txt"), os.path.join(dst, "frame_timeline.txt"))
shutil.copy(os.path.join(src, "frame_timeline_one_frame.txt"), os.path.join(dst, "frame_timeline_one_frame.txt"))
shutil.copy(os.path.join(src, "configs.txt"), os.path.join(dst, "baseline_configs.txt"))

# ---- PMC passes per kernel
pmc = {}
for kernel, name in (("k_trace_lane<false|k_trace_lane_wide|k_trace_packet", "pmc_k_trace_lane.json"), ("k_march<false", "pmc_k_march.json"), ("k_shade<false", "pmc_k_shade.json")):
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
           "tcp_lane_accesses_per_cycle_per_cu": p.get("TCP_TOTAL_CACHE_ACCESSES_sum@narrow", p["TCP_TOTAL_CACHE_ACCESSES_sum"]) / 256 / cu,
           "tcp_lane_accesses_as_run": p["TCP_TOTAL_CACHE_ACCESSES_sum"], "tcp_lane_accesses_four_wavefront_form": p.get("TCP_TOTAL_CACHE_ACCESSES_sum@narrow"),
           "l1_hit_rate": 1 - p["TCP_TCC_READ_REQ_sum"] / p["TCP_TOTAL_CACHE_ACCESSES_sum"], "l2_hit_rate": p["TCC_HIT_sum"] / p["TCC_REQ_sum"],
           "fabric_bytes_per_launch": (2 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024}
    d["derived"] = der
    json.dump(d, open(os.path.join(dst, name), "w"), indent=1)
    pmc[name] = der

# ---- README
rnd = os.path.basename(os.path.normpath(dst))
cal = json.load(open(os.path.join(root, "profiles", "round4", "valu_roof.json")))      # re-measured in round 4 on the node step the walk runs today
mix = [r for r in cal["results"] if r["class"].startswith("BVH4 LANE node-step mix") and r["waves_per_simd"] == 4][0]
old_mix = [r for r in cal["results"] if r["class"].startswith("BVH4 node-step mix") and r["waves_per_simd"] == 5][0]
tcp = json.load(open(os.path.join(root, "profiles", "round4", "tcp_access_cost.json")))
cost = tcp["min_cycles_per_counted_access"]

ks = {}
for row in csv.DictReader(open(os.path.join(dst, "kernel_stats.csv"))):
    ks[row["Name"].split("(")[0].replace("void mcrt::", "")] = row
# the walk has two forms (k_trace_lane<false>: four wavefronts per SIMD; k_trace_lane_wide: five, taken by launches of >= 4 Mi rays): one row, launch-weighted
walk_rows = [v for k, v in ks.items() if k.replace("mcrt::", "").startswith(("k_trace_lane<false>", "k_trace_lane_wide", "k_trace_packet"))]
walk_calls = sum(int(v["Calls"]) for v in walk_rows)
ks["k_trace_lane<false>"] = {"AverageNs": sum(int(v["Calls"]) * float(v["AverageNs"]) for v in walk_rows) / max(walk_calls, 1), "Calls": walk_calls}
alone = {}; walk_alone = [0.0, 0]
for line in open(os.path.join(dst, "kernels_standalone.txt")):
    for k in ("k_march<false", "k_shade<false>"):
        if k in line and " avg " in line:
            alone[k] = float(line.split(" avg ")[1].split()[0])
    if ("k_trace_lane<false>" in line or "k_trace_lane_wide" in line or "k_trace_packet" in line) and " avg " in line and " calls " in line:
        n = int(line.split(" calls ")[1].split()[0]); walk_alone[0] += n * float(line.split(" avg ")[1].split()[0]); walk_alone[1] += n
alone["k_trace_lane<false>"] = walk_alone[0] / max(walk_alone[1], 1)
r = bench["roofline"]; t = pmc["pmc_k_trace_lane.json"]; m = pmc["pmc_k_march.json"]; s = pmc["pmc_k_shade.json"]
cb = bench["cpu_baseline"]


def pipe_ms(d):          # the time a launch's counted cache accesses need at the cheapest measured cost per access
    return d["tcp_lane_accesses_per_cycle_per_cu"] * d["kernel_busy_cycles_per_cu"] * cost / (mix["clock_ghz"] * 1e6)


def krow(name, key, d, over):
    return "| `%s` | %.0f us | %.0f us | %.0f M | %.3f (%.0f %%) | %.0f %% | %.0f %% | %.0f M = %.2f ms of the pipe | %.0f MB |" % (
        name, alone.get(key, 0), over, d["valu_instructions"] / 1e6, d["valu_ipc_per_simd"], 100 * d["valu_ipc_per_simd"] / 0.5, 100 * d["valu_lane_utilisation"],
        100 * d["wave_time_waiting_on_memory"], d["tcp_lane_accesses_per_cycle_per_cu"] * d["kernel_busy_cycles_per_cu"] * 256 / 1e6, pipe_ms(d), d["fabric_bytes_per_launch"] / 1e6)


rk = r.get("kernels") or {}
def kb(name, key):
    k = rk.get(name) or {}
    return "%s beside the rest / alone %.3f / %.3f ms per launch (dilation %.2f), VALU share of the architectural rate alone %.2f" % (
        name, k.get("ms_per_launch_overlapped") or 0, k.get("ms_per_launch_alone") or 0, k.get("dilation_beside_the_rest") or 0, k.get("valu_frac_vs_architectural_alone") or 0) if k else name + ": -"
dk = drv["roofline"].get("kernels") or {}
txt = """# profiles/@RND@ -- MI355X (gfx950), ROCm 7.2

Workload of every file unless it says otherwise: `bench.py` defaults = synthetic 1 M random triangles, 128 scan-lines x 1024 sample
paths per frame, 465 RF rows, max depth 10, one GPU, @FIF@ frames in flight per pass.  Produced by `tools/profile_round.sh` on a gpurun
box and packaged by `tools/package_profiles.py`, which also wrote this file from the files beside it.  The roof and counter calibrations
(`valu_roof.json`, `tcp_access_cost.json`, `fetch_size_calibration.json`) are round 4's: `profiles/round4/`.

| file | what |
|---|---|
| `bench_unprofiled.json` | `python bench.py`: the JSON line (live PMC passes in child processes, CPU baseline, inline parity check, the per-kernel block `roofline.kernels`) |
| `bench_driver_cmd.json` | `python bench.py --gpus 1 --steps 20 --warmup 5` (the driver's command: one 20-frame pass per timed region) |
| `pmc_bench.json` | the PMC block of `bench_unprofiled.json`, the labelled fall-back `bench.py` reads when it cannot profile itself (N > 1) |
| `kernel_stats.csv` | `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps @FIF@ --warmup @FIF@ --no-cpu-baseline --no-latency-leg --no-pmc` (kernels overlap as in production) |
| `kernels_standalone.txt` | the same with `MCRT_TUNING=1 MCRT_NO_OVERLAP=1`: every kernel alone on the GPU |
| `pmc_k_trace_lane.json`, `pmc_k_march.json`, `pmc_k_shade.json` | separate `--pmc` passes (tools/pmc.sh), per-launch averages + derived figures; the walk = `k_trace_lane<false>`, `k_trace_lane_wide` and bounce 1's `k_trace_packet`, pooled |
| `frame_timeline.txt`, `frame_timeline_one_frame.txt` | start / duration of every launch of one pass: @FIF@ frames in flight, and one frame at a time |
| `baseline_configs.txt` | the five BASELINE.json configurations on one GPU (tools/configs.sh) |
| `bench_random16m.json` | `python bench.py --workload random16m`: the STREAMING regime (16 M triangles, 2 GB of BVH past the Infinity Cache) |
| `group_bench.json` | `tools/group_bench.py 0,0`: whole B-mode frames through `mcrt_group_*` (two ranks sharing the GPU) against one context |
| `bench_driver_cmd_k_path.json`, `bench_driver_cmd_wide.json`, `bench_driver_cmd_final.json` | the driver's command at three milestones of the round (the last: one frame at a time 0.872 ms): with `k_path` (one frame at a time 1.48 -> 1.11 ms), and with the kernels built without machine LICM + the five-wavefront walk from the first ray (0.386 -> 0.3595 ms per frame, one frame at a time 0.915 ms) |
| `march_layout_count_random1m.json`, `exp_march_layout.txt` | `tools/march_layout_count.py` (CPU): 128-byte lines and 4-KiB regions `k_march`'s gathers touch for six device layouts of the 256^3 texture; the layouts built and measured (all slower: recorded loss) |
| `packet_count_random1m_segdec.json`, `exp_packet_sorted.txt` | `tools/packet_count.py ... segdec` (CPU): packets over scan-lines partitioned by their rays' decisions; why sorted queues for bounces 2-3 were not built |
| `exp_path.txt`, `pmc_k_path.txt`, `frame_timeline_one_frame.txt` | the latency form `k_path`: every step measured, its counters (`tools/pmc_path.sh`), the launches of one frame |
| `frame_timeline_pass20.txt` | the driver's 20-frame pass launch by launch |
| `exp_valu_classes.txt`, `valu_classes.json`, `vgpr_bank.json`, `cndmask_cost.json`, `issue_pairs.json` | what a wave64 instruction costs gfx950 to issue: by class and operand kind, in mixed streams, and what followed from it (the build flags, `k_march`'s constants, the walk's plane picks as logic / `v_bitop3`: measured, not kept) |
| `soak_final.txt` | `tools/soak.py` on the round's final binary: 1 123 638 frames in passes of 1 / 3 / 20 / 128, every repeat identical to the first |

## What the kernels do (per launch = one bounce of a @FIF@-frame pass)

| kernel | alone | overlapped | VALU instr. | IPC / SIMD (of the 0.5 architectural; %.3f = the calibrated node-step mix) | lanes active | waiting on memory | cache accesses | fabric bytes |
|---|---|---|---|---|---|---|---|---|
%s
%s
%s

(`alone` from `kernels_standalone.txt`, `overlapped` from `kernel_stats.csv`; the walk row pools the lane walk's launches and bounce 1's packet launch.)  The BVH is served on-die: L1 hit rate
%.0f %%, L2 %.0f %% of the rest, fabric traffic %.0f MB per launch -- `roofline.hbm.measured_frac` = %.3f of the 8 TB/s HBM figure (the algorithmic
bytes, %.1f GB per launch, flow at %.1f TB/s from the caches).

`bench_unprofiled.json` (a step is a WHOLE B-mode frame: trace, accumulate, PSF, envelope, scan conversion): **%.1f M rays/s, %.3f ms per frame (%.0f frames/s)**, timed region repeated %d times (min / median / max
%.3f / %.3f / %.3f ms per frame); one frame at a time %.2f ms per frame; `roofline.frac` = %.2f of the architectural VALU issue rate (%.2f at lane level), `calibrated.frac` %.2f,
`second_roof.frac` %.2f; %s; %s.
`parity_check.rf_bit_exact` = %s on %d scan-lines.  CPU baseline (the oracle, %d usable cores of %d hardware threads, %.1f kept busy):
%.2f M rays/s, one thread %.1f k rays/s.

`bench_driver_cmd.json` (one 20-frame pass): **%.1f M rays/s, %.3f ms per frame**; the same pass with a different probe pose in every frame (`sweep`): %.3f ms per frame; %s; %s.
""" % (
    mix["simd_ipc"],
    krow("walk (k_trace_lane / _wide / _packet)", "k_trace_lane<false>", t, float(ks["k_trace_lane<false>"]["AverageNs"]) / 1e3),
    krow("k_march", "k_march<false", m, float(next(v for k, v in ks.items() if k.startswith("k_march<false, 2"))["AverageNs"]) / 1e3),
    krow("k_shade", "k_shade<false>", s, float(ks["k_shade<false>"]["AverageNs"]) / 1e3),
    100 * t["l1_hit_rate"], 100 * t["l2_hit_rate"], t["fabric_bytes_per_launch"] / 1e6, r.get("hbm_measured_frac") or 0.0, r["algorithmic_bytes_per_launch"] / 1e9, r["algorithmic_GBps_cache_served"] / 1e3,
    bench["value"] / 1e6, bench["ms_per_step"], bench["frames_per_sec"], bench["config"]["timed_region_repeats"], *bench["config"]["repeat_ms_per_step_min_median_max"],
    bench["one_frame_at_a_time"]["ms_per_step"], r.get("frac") or 0.0, r.get("frac_lane_level") or 0.0, (r.get("calibrated") or {}).get("frac") or 0.0, (r.get("second_roof") or {}).get("frac") or 0.0,
    kb("k_march", "k_march"), kb("k_shade", "k_shade"),
    bench["parity_check"]["rf_bit_exact"], bench["parity_check"]["scan_lines"], cb["cores"], cb["host"]["cpu_count"], cb["cores_kept_busy"], cb["value"] / 1e6, cb["single_thread"]["value"] / 1e3,
    drv["value"] / 1e6, drv["ms_per_step"], drv["sweep"]["ms_per_step"],
    ("k_march " + kb("", "")[2:]) if False else ("k_march beside the walk / alone %.3f / %.3f ms per launch" % ((dk.get("k_march") or {}).get("ms_per_launch_overlapped") or 0, (dk.get("k_march") or {}).get("ms_per_launch_alone") or 0)),
    "the walk %.3f / %.3f ms per launch" % ((dk.get("k_trace_lane") or {}).get("ms_per_launch_overlapped") or 0, (dk.get("k_trace_lane") or {}).get("ms_per_launch_alone") or 0))
txt = txt.replace("@FIF@", str(FIF)).replace("@RND@", rnd)
open(os.path.join(dst, "README.md"), "w").write(txt)
for extra in ("bench_random16m.json", "group_bench.json", "frame_timeline_pass20.txt", "pmc_k_path.txt"):
    if os.path.exists(os.path.join(src, extra)):
        shutil.copy(os.path.join(src, extra), os.path.join(dst, extra))
print(txt)
