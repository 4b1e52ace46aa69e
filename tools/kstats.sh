#!/bin/bash
# per-kernel totals of one bench run under rocprofv3 --kernel-trace --stats (kernels overlap as in production unless MCRT_NO_OVERLAP=1)
out=gpurun_out/kstats_$1; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/s -- python3 bench.py --steps 128 --warmup 128 --no-cpu-baseline --no-latency-leg --no-pmc ${BENCH_ARGS} > $out/bench.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/s/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage']) > 0.5: print("%-60s calls %4s avg %9.1f us  total %8.2f ms" % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
grep -h '^{' $out/bench.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.4e  ms/step %.3f' % (d['value'], d['ms_per_step']))"
