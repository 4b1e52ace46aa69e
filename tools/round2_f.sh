#!/bin/bash
# gated GPU call: fma_mix node decode, refill / leaf-batch variants, two half-GPU scan-line groups, smaller march blocks
mkdir -p gpurun_out/r2f; export TMPDIR=/tmp
O=gpurun_out/r2f
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
( time timeout 240 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -x -q -k "c1_sphere or c2_sphere or pipelines_and_walks or refit or headline" ) > $O/gate_pytest.log 2>&1 || { echo "GATE pytest failed"; tail -40 $O/gate_pytest.log | cut -c1-200; exit 1; }
tail -2 $O/gate_pytest.log
( timeout 150 python bench.py $B ) > $O/base.json 2> $O/base.err || { echo "GATE bench failed"; tail -5 $O/base.err; exit 1; }
for v in nomix r8 r32 l12 l32; do ( MCRT_LIB=$PWD/mcray-tracing_amd/build/libmcrt_hip_$v.so timeout 150 python bench.py $B ) > $O/v_$v.json 2> $O/v_$v.err; done
for tb in 640 768 1024; do ( MCRT_GROUPS=2 MCRT_TRACE_BLOCKS=$tb timeout 150 python bench.py $B ) > $O/g2_tb$tb.json 2> $O/g2_tb$tb.err; done
for mb in 1024 2048 8192 16384; do ( MCRT_MARCH_BLOCKS=$mb timeout 150 python bench.py $B ) > $O/mb$mb.json 2> $O/mb$mb.err; done
( MCRT_NO_PRIORITY=1 timeout 150 python bench.py $B ) > $O/noprio.json 2> $O/noprio.err
( MCRT_NO_OVERLAP=1 timeout 150 python bench.py $B ) > $O/nooverlap.json 2> $O/nooverlap.err
( timeout 150 python bench.py $B --steps 20 --warmup 5 ) > $O/k20.json 2> $O/k20.err
( MCRT_MARCH_BLOCKS=16384 timeout 150 python bench.py $B --steps 20 --warmup 5 ) > $O/k20_mb16384.json 2> $O/k20_mb16384.err
show $O/*.json
bash tools/pmc_quick.sh r2f > $O/pmcq.txt 2>&1; cat $O/pmcq.txt
bash tools/timeline.sh r2f > $O/timeline.txt 2>&1; head -40 $O/timeline.txt
