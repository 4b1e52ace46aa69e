mkdir -p gpurun_out/r5
export MCRT_TUNING=1
MCRT_LIB=$PWD/mcray-tracing_amd/build/libmcrt_hip_c256g16.so timeout 600 python -m pytest tests/test_gpu_baseline_configs.py -x -q -k "headline" 2>&1 | tail -2
SKIP_TESTS=1 BENCH_STEPS=20 bash tools/tune.sh base c64g8:LIB=c64g8 c64g32:LIB=c64g32 c256g16:LIB=c256g16 c1kg16:LIB=c1kg16 c4kg16:LIB=c4kg16 c1kg64:LIB=c1kg64 base2 > gpurun_out/r5/tune_ilvchunk20.txt 2>&1
SKIP_TESTS=1 BENCH_STEPS=128 bash tools/tune.sh base c64g8:LIB=c64g8 c256g16:LIB=c256g16 c1kg16:LIB=c1kg16 > gpurun_out/r5/tune_ilvchunk128.txt 2>&1
cat gpurun_out/r5/tune_ilvchunk20.txt gpurun_out/r5/tune_ilvchunk128.txt
