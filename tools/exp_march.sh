#!/bin/bash
# round 6, k_march experiments: for each library variant (name = build/libmcrt_hip_<name>.so, "default" = the in-tree library) a parity test on the
# reference's 256^3 texture, then bench.py on the driver's 20-frame pass and at 128 frames in flight with the per-kernel leg (k_march / k_shade / walk
# beside each other and alone).      tools/exp_march.sh default r5 tex1 tex2 tex4    -> gpurun_out/exp_march/summary.txt
out=gpurun_out/exp_march; mkdir -p $out
for name in "$@"; do
  lib=""; [ "$name" != default ] && lib="MCRT_LIB=$PWD/mcray-tracing_amd/build/libmcrt_hip_$name.so"
  if [ -z "$SKIP_TESTS" ]; then
  env MCRT_TUNING=1 $lib timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 240 -k "${TESTS:-headline or fast_paths or randomised}" > $out/pytest_$name.log 2>&1
  echo "$name: pytest $(tail -1 $out/pytest_$name.log)" | tee -a $out/summary.txt
  fi
  for steps in ${STEPS:-20 128}; do
    env MCRT_TUNING=1 $lib timeout 200 python bench.py --steps $steps --warmup $steps --no-cpu-baseline --no-pmc > $out/bench_${name}_$steps.log 2>&1
    python3 - $out/bench_${name}_$steps.log "$name" $steps <<'PY' | tee -a $out/summary.txt
import json, sys
try:
    d = json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][-1]); k = d['roofline']['kernels']
    f = lambda n: "%s %.4f/%.4f" % (n, k[n]['ms_per_launch_overlapped'], k[n]['ms_per_launch_alone'])
    print("%-10s steps %3s  ms/step %.4f (min %.4f)  %.4e rays/s | per launch beside/alone ms: %s  %s  %s | one frame %.3f ms" % (sys.argv[2], sys.argv[3], d['ms_per_step'], d['config']['repeat_ms_per_step_min_median_max'][0], d['value'], f('k_trace_lane'), f('k_march'), f('k_shade'), d.get('one_frame_at_a_time', {}).get('ms_per_step', 0)))
except Exception as e:
    print(sys.argv[2], sys.argv[3], 'FAILED', e)
PY
  done
done
