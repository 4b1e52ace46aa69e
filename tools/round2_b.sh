#!/bin/bash
# GPU call: full GPU test suite (fused pipeline + lane walk default), fetch-shape calibration, bench in the pipeline variants, F=1 timeline
mkdir -p gpurun_out/r2b; export TMPDIR=/tmp
O=gpurun_out/r2b
( time timeout 300 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 300 build/fetch_roof > $O/fetch_roof.json 2> $O/fetch_roof.err
( time timeout 1700 python -m pytest tests -m gpu -x -q --durations=25 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
B="--no-pmc --no-cpu-baseline"
( timeout 300 python bench.py $B ) > $O/bench_auto.json 2> $O/bench_auto.err
( MCRT_PIPELINE=fused timeout 300 python bench.py $B ) > $O/bench_fused.json 2> $O/bench_fused.err
( MCRT_PIPELINE=fused MCRT_FUSED_GROUPS=1 timeout 300 python bench.py $B ) > $O/bench_fused_g1.json 2> $O/bench_fused_g1.err
( MCRT_PIPELINE=fused MCRT_FUSED_GROUPS=4 timeout 300 python bench.py $B ) > $O/bench_fused_g4.json 2> $O/bench_fused_g4.err
( MCRT_PIPELINE=wavefront timeout 300 python bench.py $B ) > $O/bench_wave.json 2> $O/bench_wave.err
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kstats_f1 -- python3 $GRAFT_REPO_ROOT/bench.py $B --frames-in-flight 1 --steps 16 --warmup 4 --no-latency-leg > $GRAFT_REPO_ROOT/$O/kstats_f1.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kstats_f32 -- python3 $GRAFT_REPO_ROOT/bench.py $B --no-latency-leg > $GRAFT_REPO_ROOT/$O/kstats_f32.log 2>&1 )
tail -3 $O/smoke.log; tail -5 $O/pytest.log
for f in bench_auto bench_fused bench_fused_g1 bench_fused_g4 bench_wave; do python3 - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[1], 'value %.4e ms/step %.3f k_ms %.3f one-frame %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('one_frame_at_a_time',{}).get('ms_per_step')))
except Exception as e: print(sys.argv[1], 'no json', e)
PY
done
for d in kstats_f1 kstats_f32; do python3 - $O/$d <<'PY'
import csv, glob, sys
fs = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
print(sys.argv[1])
for r in csv.DictReader(open(fs[0])):
    if float(r['Percentage']) > 0.5: print("  %-70s calls %5s avg %9.1f us  total %8.2f ms" % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
done
