#!/bin/bash
# gated GPU call: subtree donation at the end of a launch / one frame at a time
mkdir -p gpurun_out/r2g; export TMPDIR=/tmp
O=gpurun_out/r2g
B="--no-pmc --no-cpu-baseline"
show() { python3 - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print("%-44s value %.4e ms/step %.3f k_ms %.3f one-frame %s" % (f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('one_frame_at_a_time',{}).get('ms_per_step')))
    except Exception as e: print(f, 'no json', e)
PY
}
( time timeout 120 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1 || { echo "GATE smoke failed"; tail -5 $O/smoke.log; exit 1; }
( time timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -x -q -k "c1_sphere or c2_sphere or pipelines_and_walks or refit or headline or randomised or liver_like or soup" ) > $O/gate_pytest.log 2>&1 || { echo "GATE pytest failed"; tail -40 $O/gate_pytest.log | cut -c1-200; exit 1; }
tail -2 $O/gate_pytest.log
( timeout 150 python bench.py $B ) > $O/base.json 2> $O/base.err || { echo "GATE bench failed"; tail -5 $O/base.err; exit 1; }
( timeout 150 python bench.py $B --steps 20 --warmup 5 ) > $O/k20.json 2> $O/k20.err
( timeout 150 python bench.py $B --frames-in-flight 1 --steps 32 --warmup 8 --no-latency-leg ) > $O/f1.json 2> $O/f1.err
( MCRT_KSPLIT_LIMIT=0 timeout 150 python bench.py $B --frames-in-flight 1 --steps 32 --warmup 8 --no-latency-leg ) > $O/f1_k0.json 2> $O/f1_k0.err
( MCRT_KSPLIT_LIMIT=0 timeout 150 python bench.py $B --steps 20 --warmup 5 ) > $O/k20_k0.json 2> $O/k20_k0.err
( timeout 150 python bench.py $B --frames-in-flight 2 --steps 32 --warmup 8 --no-latency-leg ) > $O/f2.json 2> $O/f2.err
( timeout 150 python bench.py $B --frames-in-flight 4 --steps 32 --warmup 8 --no-latency-leg ) > $O/f4.json 2> $O/f4.err
( timeout 150 python bench.py $B --frames-in-flight 8 --steps 32 --warmup 8 --no-latency-leg ) > $O/f8.json 2> $O/f8.err
show $O/*.json
BENCH_ARGS="--frames-in-flight 1" bash tools/timeline.sh r2g_f1 > $O/timeline_f1.txt 2>&1; head -36 $O/timeline_f1.txt
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log | cut -c1-200
