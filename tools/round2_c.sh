#!/bin/bash
# GPU call with gates: every step has a short timeout, and a failed gate stops the script (no more slow commands)
mkdir -p gpurun_out/r2c; export TMPDIR=/tmp
O=gpurun_out/r2c
B="--no-pmc --no-cpu-baseline"
show() { python3 - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, 'value %.4e ms/step %.3f k_ms %.3f one-frame %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('one_frame_at_a_time',{}).get('ms_per_step')))
    except Exception as e: print(f, 'no json', e)
PY
}
( time timeout 120 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1 || { echo "GATE smoke failed"; tail -5 $O/smoke.log; exit 1; }
( time timeout 240 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c1_sphere or pipelines_and_walks or frames_in_flight or large_passes" ) > $O/gate_pytest.log 2>&1 || { echo "GATE pytest failed"; tail -30 $O/gate_pytest.log; exit 1; }
( MCRT_PIPELINE=wavefront timeout 150 python bench.py $B --steps 32 --warmup 8 ) > $O/bench_wave.json 2> $O/bench_wave.err || { echo "GATE bench wavefront failed"; tail -5 $O/bench_wave.err; exit 1; }
show $O/bench_wave.json
( timeout 150 python bench.py $B --steps 32 --warmup 8 ) > $O/bench_auto.json 2> $O/bench_auto.err || { echo "GATE bench auto failed (fused latency leg)"; tail -5 $O/bench_auto.err; exit 1; }
show $O/bench_auto.json
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
for g in 1 2 4; do ( MCRT_PIPELINE=fused MCRT_FUSED_GROUPS=$g timeout 150 python bench.py $B --frames-in-flight 1 --steps 32 --warmup 8 --no-latency-leg ) > $O/bench_f1_fused_g$g.json 2> $O/bench_f1_fused_g$g.err; done
( MCRT_PIPELINE=fused MCRT_FUSED_GROUPS=2 MCRT_TRACE_BLOCKS=1024 timeout 150 python bench.py $B --frames-in-flight 1 --steps 32 --warmup 8 --no-latency-leg ) > $O/bench_f1_fused_g2_b1024.json 2> $O/bench_f1_fused_g2_b1024.err
( MCRT_PIPELINE=wavefront timeout 150 python bench.py $B --frames-in-flight 1 --steps 32 --warmup 8 --no-latency-leg ) > $O/bench_f1_wave.json 2> $O/bench_f1_wave.err
( MCRT_PIPELINE=wavefront MCRT_QUAD_WALK=1 timeout 150 python bench.py $B --frames-in-flight 1 --steps 32 --warmup 8 --no-latency-leg ) > $O/bench_f1_quad.json 2> $O/bench_f1_quad.err
( MCRT_PIPELINE=fused timeout 150 python bench.py $B --no-latency-leg ) > $O/bench_f32_fused.json 2> $O/bench_f32_fused.err
show $O/bench_f1_fused_g1.json $O/bench_f1_fused_g2.json $O/bench_f1_fused_g4.json $O/bench_f1_fused_g2_b1024.json $O/bench_f1_wave.json $O/bench_f1_quad.json $O/bench_f32_fused.json
ks() { ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/$1 -- python3 $GRAFT_REPO_ROOT/bench.py $B --no-latency-leg $2 > $GRAFT_REPO_ROOT/$O/$1.log 2>&1 ); python3 - $O/$1 <<'PY'
import csv, glob, sys
fs = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
print(sys.argv[1])
for r in csv.DictReader(open(fs[0])):
    if float(r['Percentage']) > 0.5: print("  %-70s calls %5s avg %9.1f us  total %8.2f ms" % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
}
ks kstats_f32 ""
MCRT_NO_OVERLAP=1 ks kstats_f32_alone ""
ks kstats_f1 "--frames-in-flight 1 --steps 16 --warmup 4"
( timeout 400 python bench.py ) > $O/bench_full.json 2> $O/bench_full.err; show $O/bench_full.json; tail -3 $O/bench_full.err
