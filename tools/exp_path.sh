#!/bin/bash
# round 6: the latency form (k_path: one launch carries every path through all of its bounces) against the staged pipeline, by frames per pass.
#     tools/exp_path.sh [frames-in-flight ...]     -> gpurun_out/exp_path/summary.txt   (MCRT_PATH_MAX: 0 = staged, 100000000 = k_path at any size)
out=gpurun_out/exp_path; mkdir -p $out
for F in "${@:-1 2 3 4}"; do
  for pm in 0 100000000; do
    env MCRT_TUNING=1 MCRT_PATH_MAX=$pm timeout 200 python bench.py --steps 24 --warmup 24 --frames-in-flight $F --no-cpu-baseline --no-pmc --no-latency-leg ${BENCH_ARGS} > $out/bench_${F}_$pm.log 2>&1
    python3 - $out/bench_${F}_$pm.log $F $pm <<'PY' | tee -a $out/summary.txt
import json, sys
try:
    d = json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][-1])
    print("frames per pass %2s  %-8s ms/frame %.4f (min %.4f max %.4f)  %.4e rays/s  walk-kind launch %.3f ms x %.2f per frame" % (sys.argv[2], "k_path" if sys.argv[3] != "0" else "staged", d['ms_per_step'], *d['config']['repeat_ms_per_step_min_median_max'][::2], d['value'], d['roofline']['kernel_ms'], d['roofline']['launches_per_frame']))
except Exception as e:
    print(sys.argv[2], sys.argv[3], 'FAILED', e)
PY
  done
done
