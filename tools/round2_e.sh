#!/bin/bash
# gated GPU call: half-float walk nodes
mkdir -p gpurun_out/r2e; export TMPDIR=/tmp
O=gpurun_out/r2e
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
( time timeout 240 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c1_sphere or c2_sphere or pipelines_and_walks or device_lbvh_gives or refit" ) > $O/gate_pytest.log 2>&1 || { echo "GATE pytest failed"; tail -40 $O/gate_pytest.log | cut -c1-200; exit 1; }
( timeout 150 python bench.py $B ) > $O/bench_gate.json 2> $O/bench_gate.err || { echo "GATE bench failed"; tail -5 $O/bench_gate.err; exit 1; }
show $O/bench_gate.json
( time timeout 900 python -m pytest tests -m gpu -q --durations=5 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log | cut -c1-200
( timeout 150 python bench.py $B --steps 20 --warmup 5 ) > $O/k20.json 2> $O/k20.err
( timeout 150 python bench.py $B --frames-in-flight 64 --steps 128 --no-latency-leg ) > $O/f64.json 2> $O/f64.err
( timeout 150 python bench.py $B --frames-in-flight 48 --steps 96 --no-latency-leg ) > $O/f48.json 2> $O/f48.err
( timeout 150 python bench.py $B --frames-in-flight 16 --steps 64 --no-latency-leg ) > $O/f16.json 2> $O/f16.err
( MCRT_QUAD_WALK=1 timeout 150 python bench.py $B --steps 20 --warmup 5 ) > $O/k20_quad.json 2> $O/k20_quad.err
show $O/k20.json $O/f64.json $O/f48.json $O/f16.json $O/k20_quad.json
MCRT_NO_OVERLAP=1 bash tools/kstats.sh r2e_alone > $O/kstats_alone.txt 2>&1; cat $O/kstats_alone.txt
bash tools/pmc_quick.sh r2e > $O/pmcq.txt 2>&1; cat $O/pmcq.txt
