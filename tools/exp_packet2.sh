mkdir -p gpurun_out/r5
export MCRT_TUNING=1
for F in 1 2 4 8; do
  SKIP_TESTS=1 BENCH_STEPS=16 BENCH_ARGS="--frames-in-flight $F" bash tools/tune.sh f${F}base f${F}pk1:MCRT_PACKET_BOUNCES=2
done > gpurun_out/r5/tune_packet_small.txt 2>&1
cat gpurun_out/r5/tune_packet_small.txt
bash tools/configs.sh > gpurun_out/r5/configs_base.txt 2>&1
MCRT_PACKET_BOUNCES=2 bash tools/configs.sh > gpurun_out/r5/configs_pk1.txt 2>&1
paste -d'\n' gpurun_out/r5/configs_base.txt gpurun_out/r5/configs_pk1.txt | cut -c1-150
python bench.py --workload random16m --steps 16 --warmup 16 --no-cpu-baseline --no-latency-leg --no-pmc 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('random16m base', d['value'], d['ms_per_step'])"
MCRT_PACKET_BOUNCES=2 python bench.py --workload random16m --steps 16 --warmup 16 --no-cpu-baseline --no-latency-leg --no-pmc 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('random16m pk1 ', d['value'], d['ms_per_step'])"
