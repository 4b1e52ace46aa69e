#!/usr/bin/env python3
"""Copies the summaries of a tools/pmc.sh + tools/timeline.sh + bench.py run (gpurun_out/) into profiles/<round>/ and
derives the figures DESIGN.md quotes.  usage: package_profiles.py <tag> <round dir>   (tag as given to pmc.sh / timeline.sh)"""
import csv, glob, json, os, shutil, subprocess, sys
tag, dst = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
os.makedirs(dst, exist_ok=True)
shutil.copy(glob.glob(os.path.join(out, "pmc_" + tag, "stats", "*", "*kernel_stats.csv"))[0], os.path.join(dst, "kernel_stats.csv"))
shutil.copy(os.path.join(out, "pmc_" + tag, "stats.log"), os.path.join(dst, "bench_under_rocprof_stats.log"))
shutil.copy(os.path.join(out, "timeline_%s.txt" % tag), os.path.join(dst, "frame_timeline.txt"))
line = [l for l in open(os.path.join(out, "bench_%s.json" % tag)) if l.startswith("{")][-1]
open(os.path.join(dst, "bench_unprofiled.json"), "w").write(line)
for kernel, name in (("k_trace<false", "pmc_k_trace.json"), ("k_march<false", "pmc_k_march.json"), ("k_shade<false", "pmc_k_shade.json")):
    env = dict(os.environ, PMC_KERNEL=kernel)
    subprocess.check_output([sys.executable, os.path.join(root, "tools", "pmc_summary.py"), os.path.join(out, "pmc_" + tag)], env=env)
    d = json.load(open(os.path.join(out, "pmc_" + tag, "summary.json")))
    p = {k: v["avg_per_launch"] for k, v in d["pmc"].items()}
    der = {"note": "FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md (HBM section); "
                   "separate --pmc passes; bench.py defaults (32 frames in flight); kernels run one at a time under --pmc"}
    der["traffic_bytes_per_launch"] = (2 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024
    if kernel.startswith("k_trace"): der["traffic_bytes_per_k_trace_launch"] = der["traffic_bytes_per_launch"]
    der["l1_hit_rate"] = 1 - p["TCP_TCC_READ_REQ_sum"] / p["TCP_TOTAL_CACHE_ACCESSES_sum"]
    der["l2_hit_rate"] = p["TCC_HIT_sum"] / p["TCC_REQ_sum"]
    der["valu_lane_utilisation"] = p["SQ_THREAD_CYCLES_VALU"] / (64 * p["SQ_ACTIVE_INST_VALU"])
    der["avg_l2_read_latency_cycles"] = p["TCP_TCC_READ_REQ_LATENCY_sum"] / p["TCP_TCC_READ_REQ_sum"]
    der["kernel_cycles_per_cu"] = p["SQ_BUSY_CU_CYCLES"] / 256
    # a wave64 VALU instruction occupies its SIMD16 for 4 cycles: share of the 1024 SIMDs' cycles spent issuing VALU work
    der["valu_busy_share"] = 4 * p["SQ_INSTS_VALU"] / (1024 * der["kernel_cycles_per_cu"])
    der["valu_instructions"] = p["SQ_INSTS_VALU"]; der["salu_instructions"] = p["SQ_INSTS_SALU"]
    d["derived"] = der
    json.dump(d, open(os.path.join(dst, name), "w"), indent=1)
    print(name, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in der.items() if k != "note"})
