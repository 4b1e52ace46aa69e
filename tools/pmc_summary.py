#!/usr/bin/env python3
"""Summarise rocprofv3 output dirs (kernel stats + PMC passes) of tools/pmc.sh into one JSON + text table."""
import csv, glob, json, os, sys, collections
KERNELS = os.environ.get("PMC_KERNEL", "k_trace_lane<false>|k_trace_lane_wide|k_trace_packet").split("|")   # substrings of the kernel(s) the counters are reported for, pooled (the walk has two forms)
out = sys.argv[1]
res = {"kernel_stats": [], "pmc": {}}


def newest_only(paths):
    """rocprofv3 names its files by process id, and gpurun MERGES a run's outputs into the local copy of the directory: keep the newest run's file"""
    paths = sorted(paths, key=os.path.getmtime)
    return paths[-1:]

for f in newest_only(glob.glob(out + "/stats/*/*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.05:
            res["kernel_stats"].append({k: r[k] for k in ("Name", "Calls", "AverageNs", "Percentage", "MinNs", "MaxNs")})
for d in sorted(glob.glob(out + "/*/")):
    for f in newest_only(glob.glob(d + "*/*counter_collection.csv")):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        pooled = collections.defaultdict(list)
        for k, v in agg.items():
            if any(name in k for name in KERNELS):
                for c, x in v.items():
                    pooled[c] += x
        suffix = "@narrow" if os.path.basename(d.rstrip("/")).endswith("_narrow") else ""
        for c, x in pooled.items():
            c = c + suffix
            res["pmc"][c] = {"avg_per_launch": sum(x) / len(x), "launches": len(x), "pass": os.path.basename(d.rstrip("/"))}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for k in res["kernel_stats"]:
    print("%-90s calls %5s avg %10.1f us  %5s%%" % (k["Name"][:90], k["Calls"], float(k["AverageNs"]) / 1e3, k["Percentage"]))
for c, v in sorted(res["pmc"].items()):
    print("%-34s %16.1f  (%s, n=%d)" % (c, v["avg_per_launch"], v["pass"], v["launches"]))
