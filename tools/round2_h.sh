#!/bin/bash
# gated GPU call: helper-lane tie fix, k_march pool + row stepper
mkdir -p gpurun_out/r2h; export TMPDIR=/tmp
O=gpurun_out/r2h
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
( time timeout 600 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log | cut -c1-200
grep -q "rc=0" $O/pytest.log || { grep -E "^E  |^FAILED" $O/pytest.log | head -10 | cut -c1-300; exit 1; }
( timeout 150 python bench.py $B ) > $O/base.json 2> $O/base.err || { echo "GATE bench failed"; tail -5 $O/base.err; exit 1; }
( timeout 150 python bench.py $B --steps 20 --warmup 5 ) > $O/k20.json 2> $O/k20.err
show $O/*.json
bash tools/pmc_quick.sh r2h > $O/pmcq.txt 2>&1; cat $O/pmcq.txt
MCRT_NO_OVERLAP=1 bash tools/kstats.sh r2h_alone > $O/kstats_alone.txt 2>&1; cat $O/kstats_alone.txt
