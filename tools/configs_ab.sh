#!/bin/bash
mkdir -p gpurun_out/cfgab
run() { name=$1; lib=$2; shift; shift
  L=""; [ $lib != default ] && L="MCRT_LIB=$PWD/mcray-tracing_amd/build/libmcrt_hip_$lib.so"
  env $L timeout 600 python bench.py --no-cpu-baseline --no-latency-leg --no-pmc "$@" > gpurun_out/cfgab/${name}_$lib.log 2>&1
  python3 - gpurun_out/cfgab/${name}_$lib.log $name $lib <<'PY'
import json,sys
try:
    d=json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][-1])
    print("%-4s %-8s %.4e rays/s %9.4f ms/frame (min %.4f)" % (sys.argv[2], sys.argv[3], d['value'], d['ms_per_step'], d['config']['repeat_ms_per_step_min_median_max'][0]))
except Exception as e: print(sys.argv[2], sys.argv[3], 'FAILED', e)
PY
}
for lib in default licm default licm; do
run C2 $lib --workload sphere --scanlines 128 --rays 1024 --rows 512 --steps 128 --warmup 128
run C3 $lib --workload liver --scanlines 128 --rays 4096 --steps 32 --warmup 16 --frames-in-flight 16
run C5 $lib --workload liver --scanlines 512 --rays 16384 --steps 4 --warmup 2 --frames-in-flight 2
done
