#!/bin/bash
# gated GPU call: tests, tuning variants of the lane walk, K-split at one frame at a time, PMC passes of the lane walk and k_march
mkdir -p gpurun_out/r2d; export TMPDIR=/tmp
O=gpurun_out/r2d
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
( time timeout 240 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "contract_math or c1_sphere or pipelines_and_walks or frames_in_flight" ) > $O/gate_pytest.log 2>&1 || { echo "GATE pytest failed"; tail -30 $O/gate_pytest.log; exit 1; }
( timeout 150 python bench.py $B --steps 32 --warmup 8 ) > $O/bench_gate.json 2> $O/bench_gate.err || { echo "GATE bench failed"; tail -5 $O/bench_gate.err; exit 1; }
show $O/bench_gate.json
( time timeout 900 python -m pytest tests -m gpu -q --durations=10 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log | cut -c1-200
# ---- throughput: tuning variants (32 frames in flight)
( timeout 150 python bench.py $B --no-latency-leg ) > $O/f32_base.json 2> $O/f32_base.err
for v in s24 s16 s24r8 s24r32 s24l12 s24l32; do ( MCRT_LIB=$PWD/mcray-tracing_amd/build/libmcrt_hip_$v.so timeout 150 python bench.py $B --no-latency-leg ) > $O/f32_$v.json 2> $O/f32_$v.err; done
( MCRT_TRACE_BLOCKS=1536 MCRT_LIB=$PWD/mcray-tracing_amd/build/libmcrt_hip_s16w6.so timeout 150 python bench.py $B --no-latency-leg ) > $O/f32_s16w6.json 2> $O/f32_s16w6.err
( MCRT_TRACE_BLOCKS=1024 MCRT_LIB=$PWD/mcray-tracing_amd/build/libmcrt_hip_s24.so timeout 150 python bench.py $B --no-latency-leg ) > $O/f32_s24_b1024.json 2> $O/f32_s24_b1024.err
( MCRT_GROUPS=2 timeout 150 python bench.py $B --no-latency-leg ) > $O/f32_groups2.json 2> $O/f32_groups2.err
show $O/f32_*.json
# ---- latency: one frame at a time, K-split limits, both walks
for k in 131072 262144 524288 1048576; do
  ( MCRT_KSPLIT_LIMIT=$k timeout 120 python bench.py $B --frames-in-flight 1 --steps 32 --warmup 8 --no-latency-leg ) > $O/f1_lane_k$k.json 2> $O/f1_lane_k$k.err
  ( MCRT_QUAD_WALK=1 MCRT_KSPLIT_LIMIT=$k timeout 120 python bench.py $B --frames-in-flight 1 --steps 32 --warmup 8 --no-latency-leg ) > $O/f1_quad_k$k.json 2> $O/f1_quad_k$k.err
done
show $O/f1_*.json
# ---- timelines
BENCH_ARGS="--frames-in-flight 1" bash tools/timeline.sh r2d_f1 > $O/timeline_f1.txt 2>&1
MCRT_QUAD_WALK=1 MCRT_KSPLIT_LIMIT=1048576 BENCH_ARGS="--frames-in-flight 1" bash tools/timeline.sh r2d_f1_quad_k1m > $O/timeline_f1_quad_k1m.txt 2>&1
# ---- PMC passes (lane walk; then k_march)
bash tools/pmc.sh r2d > $O/pmc_trace.txt 2>&1
PMC_KERNEL="k_march<false" python3 tools/pmc_summary.py gpurun_out/pmc_r2d > $O/pmc_march.txt 2>&1
PMC_KERNEL="k_shade<false" python3 tools/pmc_summary.py gpurun_out/pmc_r2d > $O/pmc_shade.txt 2>&1
tail -45 $O/pmc_trace.txt
