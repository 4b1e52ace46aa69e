#!/bin/bash
out=gpurun_out/exp_path2; mkdir -p $out; rm -f $out/summary.txt
for lib in default o32w4; do for G in 1 2 4 8; do
  L=""; [ $lib != default ] && L="MCRT_LIB=$PWD/mcray-tracing_amd/build/libmcrt_hip_$lib.so"
  env MCRT_TUNING=1 MCRT_PATH_GROUPS=$G $L timeout 200 python bench.py --steps 24 --warmup 24 --frames-in-flight 1 --no-cpu-baseline --no-pmc --no-latency-leg > $out/b_${lib}_$G.log 2>&1
  python3 - $out/b_${lib}_$G.log $lib $G <<'PY' | tee -a $out/summary.txt
import json, sys
try:
    d = json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][-1])
    print("%-8s groups %s  ms/frame %.4f (min %.4f max %.4f)" % (sys.argv[2], sys.argv[3], d['ms_per_step'], *d['config']['repeat_ms_per_step_min_median_max'][::2]))
except Exception as e:
    print(sys.argv[2], sys.argv[3], 'FAILED', e)
PY
done; done
