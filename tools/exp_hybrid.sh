mkdir -p gpurun_out/r5
export MCRT_TUNING=1
MCRT_PACKET_BOUNCES=0 MCRT_HYBRID_BOUNCES=0xfffe MCRT_PACKET_FROM=0 timeout 900 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_parity.py -x -q -k "headline or c3 or randomised or c1 or reference_shape or schedules" 2>&1 | tail -3
MCRT_HYBRID_BOUNCES=4 python bench.py --steps 128 --warmup 128 --no-pmc --no-latency-leg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('128f p1h2', d['value'], d['ms_per_step'], d['parity_check']['rf_bit_exact'])"
SKIP_TESTS=1 BENCH_STEPS=20 bash tools/tune.sh base h1:MCRT_PACKET_BOUNCES=0,MCRT_HYBRID_BOUNCES=2 p1h2:MCRT_HYBRID_BOUNCES=4 p1h23:MCRT_HYBRID_BOUNCES=12 p1h2345:MCRT_HYBRID_BOUNCES=60 p1hall:MCRT_HYBRID_BOUNCES=0x3fc base2 > gpurun_out/r5/tune_hybrid20.txt 2>&1
SKIP_TESTS=1 BENCH_STEPS=128 bash tools/tune.sh base p1h2:MCRT_HYBRID_BOUNCES=4 p1h2345:MCRT_HYBRID_BOUNCES=60 > gpurun_out/r5/tune_hybrid128.txt 2>&1
cat gpurun_out/r5/tune_hybrid20.txt gpurun_out/r5/tune_hybrid128.txt
