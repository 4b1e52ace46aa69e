#!/bin/bash
# counters of k_path (the latency form) over one-frame passes of the headline workload: separate rocprofv3 --pmc passes, as the pool requires
out=gpurun_out/pmc_path; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
P="python3 tools/path_stamps.py"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $P > $out/stats.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $out/sq1 -- $P > $out/sq1.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $out/sq2 -- $P > $out/sq2.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $out/tcp -- $P > $out/tcp.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/tcc -- $P > $out/tcc.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
for f in glob.glob(d + '/stats/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r['Percentage']) > 1: print("%-70s calls %4s avg %9.1f us" % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
per = collections.defaultdict(list)
for f in glob.glob(d + '/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_path' in r['Kernel_Name']: per[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(per): print("k_path %-36s per launch %.4g (%d launches)" % (k, sum(per[k]) / len(per[k]), len(per[k])))
PY
