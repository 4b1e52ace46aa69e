#!/bin/bash
# the BASELINE.json configurations on ONE GPU (C1..C5), one line each
mkdir -p gpurun_out/configs
run() { name=$1; shift
  timeout 600 python bench.py --no-cpu-baseline --no-latency-leg --no-pmc "$@" > gpurun_out/configs/$name.log 2>&1
  python3 - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads([x for x in open('gpurun_out/configs/%s.log'%n) if x.startswith('{')][-1]); r=d['roofline']
    print("%-4s %-62s %.3e rays/s %8.2f ms/frame  walk %.0f GB/s algorithmic, cache-served (%.2f of the HBM figure)" % (n, d['config']['workload'][:62], d['value'], d['ms_per_step'], r['algorithmic_GBps_cache_served'], r['algorithmic_GBps_cache_served'] / 8000.0))
except Exception as e:
    print(n,'FAILED',e, open('gpurun_out/configs/%s.log'%n).read()[-300:])
PY
}
run C1 --workload sphere --scanlines 32 --rays 64 --steps 256 --warmup 256
run C2 --workload sphere --scanlines 128 --rays 1024 --rows 512 --steps 128 --warmup 128
run C3 --workload liver --scanlines 128 --rays 4096 --steps 32 --warmup 16 --frames-in-flight 16
run C4 --workload random1m --scanlines 256 --rays 8192 --steps 8 --warmup 4 --frames-in-flight 4
run C5 --workload liver --scanlines 512 --rays 16384 --steps 4 --warmup 2 --frames-in-flight 2
