run() { name=$1; shift; python bench.py --no-cpu-baseline --no-latency-leg --no-pmc "$@" > gpurun_out/cs_$name.log 2>&1; python3 -c "
import json,sys
try:
    d=json.loads([x for x in open('gpurun_out/cs_$name.log') if x.startswith('{')][-1]); print('%-14s %.3e rays/s %8.3f ms/frame' % ('$name', d['value'], d['ms_per_step']))
except Exception as e: print('$name FAILED', e)
"; }
export MCRT_TUNING=1
for mb in auto 4096 8192; do
  if [ $mb != auto ]; then export MCRT_MARCH_BLOCKS=$mb; else unset MCRT_MARCH_BLOCKS; fi
  run C3_$mb --workload liver --scanlines 128 --rays 4096 --steps 32 --warmup 16 --frames-in-flight 16
  run C4_$mb --workload random1m --scanlines 256 --rays 8192 --steps 8 --warmup 4 --frames-in-flight 4
  run C5_$mb --workload liver --scanlines 512 --rays 16384 --steps 4 --warmup 2 --frames-in-flight 2
  run C2_$mb --workload sphere --scanlines 128 --rays 1024 --rows 512 --steps 128 --warmup 128
  run H20_$mb --steps 20 --warmup 20
  run H1_$mb --steps 32 --warmup 32 --frames-in-flight 1
  run H4_$mb --steps 32 --warmup 32 --frames-in-flight 4
  run H8_$mb --steps 32 --warmup 32 --frames-in-flight 8
done
