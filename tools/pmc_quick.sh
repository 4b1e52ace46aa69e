#!/bin/bash
# instruction counts only (two --pmc passes) for the kernels of one bench run: quick before/after check of a kernel change
out=gpurun_out/pmcq_$1; mkdir -p $out; export TMPDIR=/tmp
B="python3 bench.py --steps 32 --warmup 32 --no-cpu-baseline --no-latency-leg --no-pmc ${BENCH_ARGS}"
timeout -s KILL 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $out/q1 -- $B > $out/q1.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/q1/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0].replace('void mcrt::', '')][r['Counter_Name']].append(float(r['Counter_Value']))
# the walk's launches of the last pass, bounce by bounce (dispatch order)
rows = []
for f in glob.glob(sys.argv[1] + '/q1/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_trace_lane<false' in r['Kernel_Name'] or 'k_trace_lane_wide' in r['Kernel_Name']: rows.append((int(r['Dispatch_Id']), r['Counter_Name'], float(r['Counter_Value'])))
per = collections.defaultdict(dict)
for d, c, v in rows: per[d][c] = per[d].get(c, 0.0) + v
for d in sorted(per)[-10:]:
    a = per[d]
    print("k_trace_lane dispatch %6d  VALU %7.1f M  busy cycles/CU %6.2f M  lane util %4.1f%%  VALU/cycle/SIMD %.3f" % (d, a['SQ_INSTS_VALU'] / 1e6, a['SQ_BUSY_CU_CYCLES'] / 256e6,
          100 * a['SQ_THREAD_CYCLES_VALU'] / (64 * a['SQ_ACTIVE_INST_VALU']), a['SQ_INSTS_VALU'] / (1024 * a['SQ_BUSY_CU_CYCLES'] / 256)))
for k, v in sorted(agg.items()):
    if '<false' not in k and 'k_trace_lane_wide' not in k: continue
    a = {c: sum(x) / len(x) for c, x in v.items()}
    print("%-16s n=%3d  VALU %7.1f M  SALU %7.1f M  VMEM_RD %6.2f M  LDS %6.2f M  busy cycles/CU %6.2f M  VALU busy %4.1f%%  lane util %4.1f%%" % (
        k, len(v['SQ_INSTS_VALU']), a['SQ_INSTS_VALU'] / 1e6, a['SQ_INSTS_SALU'] / 1e6, a['SQ_INSTS_VMEM_RD'] / 1e6, a['SQ_INSTS_LDS'] / 1e6, a['SQ_BUSY_CU_CYCLES'] / 256e6,
        100 * 4 * a['SQ_INSTS_VALU'] / (1024 * a['SQ_BUSY_CU_CYCLES'] / 256), 100 * a['SQ_THREAD_CYCLES_VALU'] / (64 * a['SQ_ACTIVE_INST_VALU'])))
PY
