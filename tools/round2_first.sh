#!/bin/bash
# GPU call: smoke, VALU / fetch calibration (+ PMC cross-check), GPU test suite, bench A/B lane walk vs quad walk
mkdir -p gpurun_out/r2a; export TMPDIR=/tmp
O=gpurun_out/r2a
( time timeout 600 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 300 build/valu_roof > $O/valu_roof.json 2> $O/valu_roof.err
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $GRAFT_REPO_ROOT/$O/valu_pmc -- $GRAFT_REPO_ROOT/build/valu_roof --quick > $GRAFT_REPO_ROOT/$O/valu_pmc.json 2> $GRAFT_REPO_ROOT/$O/valu_pmc.err )
( time timeout 300 python bench.py --no-pmc --no-cpu-baseline ) > $O/bench_lane.json 2> $O/bench_lane.err
( time MCRT_QUAD_WALK=1 timeout 300 python bench.py --no-pmc --no-cpu-baseline ) > $O/bench_quad.json 2> $O/bench_quad.err
( time timeout 1700 python -m pytest tests -m gpu -x -q --durations=25 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
( time timeout 600 python bench.py ) > $O/bench.json 2> $O/bench.err
( time timeout 300 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline ) > $O/bench_k20.json 2> $O/bench_k20.err
tail -3 $O/smoke.log; tail -5 $O/pytest.log; for f in bench_lane bench_quad bench bench_k20; do python3 - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[1], 'value %.4e ms/step %.3f k_ms %.3f one-frame %s parity %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('one_frame_at_a_time',{}).get('ms_per_step'), d.get('parity_check')))
except Exception as e: print(sys.argv[1], 'no json', e)
PY
done
