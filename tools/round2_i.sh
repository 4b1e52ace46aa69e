#!/bin/bash
# gated GPU call: k_march variants, residency cap of k_march (LDS padding)
mkdir -p gpurun_out/r2i; export TMPDIR=/tmp
O=gpurun_out/r2i
B="--no-pmc --no-cpu-baseline --no-latency-leg"
show() { python3 - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print("%-44s value %.4e ms/step %.3f k_ms %.3f" % (f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))
    except Exception as e: print(f, 'no json', e)
PY
}
( time timeout 120 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1 || { echo "GATE smoke failed"; tail -5 $O/smoke.log; exit 1; }
( time timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c1_sphere or c2_sphere or pipelines_and_walks or reference_shape or randomised" ) > $O/gate_pytest.log 2>&1 || { echo "GATE pytest failed"; tail -30 $O/gate_pytest.log | cut -c1-200; exit 1; }
tail -1 $O/gate_pytest.log
( timeout 150 python bench.py $B ) > $O/base.json 2> $O/base.err || { echo "GATE bench failed"; tail -5 $O/base.err; exit 1; }
( timeout 150 python bench.py $B --steps 20 --warmup 5 ) > $O/k20.json 2> $O/k20.err
for v in pool norowfast refill2 refill8; do ( MCRT_LIB=$PWD/mcray-tracing_amd/build/libmcrt_hip_$v.so timeout 150 python bench.py $B ) > $O/v_$v.json 2> $O/v_$v.err; done
for pad in 6000 12000 18000 24000 32000; do ( MCRT_MARCH_LDS_PAD=$pad timeout 150 python bench.py $B ) > $O/pad$pad.json 2> $O/pad$pad.err; ( MCRT_MARCH_LDS_PAD=$pad timeout 150 python bench.py $B --steps 20 --warmup 5 ) > $O/k20_pad$pad.json 2> $O/k20_pad$pad.err; done
show $O/*.json
MCRT_NO_OVERLAP=1 bash tools/kstats.sh r2i_alone > $O/kstats_alone.txt 2>&1; head -4 $O/kstats_alone.txt
