#!/bin/bash
# throughput vs problem size (rays per scan-line) on one GPU
mkdir -p gpurun_out/scale
for r in "$@"; do
  timeout 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pmc --rays $r > gpurun_out/scale/r$r.log 2>&1
  python3 - $r <<'PY'
import json,sys
r=sys.argv[1]
try:
    l=[x for x in open('gpurun_out/scale/r%s.log'%r) if x.startswith('{')][-1]; d=json.loads(l); ro=d['roofline']
    print("rays %6s  value %.3e rays/s  ms/step %8.3f  k_trace %.3f ms/frame  algorithmic %.0f GB/s" % (r, d['value'], d['ms_per_step'], ro['kernel_ms']*ro.get('launches_per_frame',1), ro['algorithmic_GBps_cache_served']))
except Exception as e:
    print(r,'FAILED',e, open('gpurun_out/scale/r%s.log'%r).read()[-400:])
PY
done
