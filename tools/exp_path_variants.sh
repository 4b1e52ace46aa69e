#!/bin/bash
# one frame at a time (bench.py --frames-in-flight 1) per library variant: tools/exp_path_variants.sh name...   -> gpurun_out/exp_path_variants/summary.txt
out=gpurun_out/exp_path_variants; mkdir -p $out
for lib in "$@"; do
  L=""; [ $lib != default ] && L="MCRT_LIB=$PWD/mcray-tracing_amd/build/libmcrt_hip_$lib.so"
  env MCRT_TUNING=1 $L timeout 200 python bench.py --steps 24 --warmup 24 --frames-in-flight 1 --no-cpu-baseline --no-pmc --no-latency-leg > $out/b_${lib}.log 2>&1
  python3 - $out/b_${lib}.log $lib <<'PY' | tee -a $out/summary.txt
import json, sys
try:
    d = json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][-1])
    print("%-14s one frame at a time  ms/frame %.4f (min %.4f max %.4f)  k_path launch %.3f ms" % (sys.argv[2], d['ms_per_step'], *d['config']['repeat_ms_per_step_min_median_max'][::2], d['roofline']['kernel_ms']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
