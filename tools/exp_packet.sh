mkdir -p gpurun_out/r5
export MCRT_TUNING=1
MCRT_PACKET_BOUNCES=0x3fe timeout 900 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_parity.py -x -q -k "headline or c3 or randomised or c1 or reference_shape" 2>&1 | tail -3
SKIP_TESTS=1 BENCH_STEPS=20 bash tools/tune.sh base pk1:MCRT_PACKET_BOUNCES=2 pk12:MCRT_PACKET_BOUNCES=6 base2 > gpurun_out/r5/tune_packet20.txt 2>&1
SKIP_TESTS=1 BENCH_STEPS=128 bash tools/tune.sh base pk1:MCRT_PACKET_BOUNCES=2 pk12:MCRT_PACKET_BOUNCES=6 > gpurun_out/r5/tune_packet128.txt 2>&1
cat gpurun_out/r5/tune_packet20.txt gpurun_out/r5/tune_packet128.txt
