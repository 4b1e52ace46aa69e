#!/bin/bash
# tuning sweep on the GPU box: parity first, then bench.py per spec.  spec = name[:VAR=VAL[,VAR=VAL...]]
# (VAR LIB=<variant> selects mcray-tracing_amd/build/libmcrt_hip_<variant>.so)
mkdir -p gpurun_out/tune
if [ -z "$SKIP_TESTS" ]; then
(timeout 600 python -m pytest tests -m gpu -x -q --timeout 240 > gpurun_out/tune/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/tune/pytest.log); tail -4 gpurun_out/tune/pytest.log
fi
for spec in "$@"; do
  name=${spec%%:*}; vars=""
  if [[ "$spec" == *:* ]]; then vars=$(echo "${spec#*:}" | tr ',' ' '); fi
  envs="MCRT_TUNING=1"      # (the knobs are only read under MCRT_TUNING=1)
  for v in $vars; do
    if [[ "$v" == LIB=* ]]; then envs="$envs MCRT_LIB=$PWD/mcray-tracing_amd/build/libmcrt_hip_${v#LIB=}.so"; else envs="$envs $v"; fi
  done
  env $envs timeout 90 python bench.py --steps ${BENCH_STEPS:-128} --warmup ${BENCH_STEPS:-128} --no-cpu-baseline --no-latency-leg --no-pmc ${BENCH_ARGS} > gpurun_out/tune/$name.log 2>&1
  python3 - "$name" <<'PY'
import json,sys
name=sys.argv[1]
try:
    l=[x for x in open('gpurun_out/tune/%s.log'%name) if x.startswith('{')][-1]; d=json.loads(l)
    r=d['roofline']
    print("%-16s value %.3e rays/s  ms/step %.3f  k_trace avg %.4f ms x%d = %.3f ms/frame  algorithmic %.0f GB/s" % (name, d['value'], d['ms_per_step'], r['kernel_ms'], int(round(r.get('launches_per_frame',1))), r['kernel_ms']*r.get('launches_per_frame',1), r["algorithmic_GBps_cache_served"]))
except Exception as e:
    print(name, 'FAILED', e)
PY
done
