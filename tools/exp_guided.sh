mkdir -p gpurun_out/r5
export MCRT_TUNING=1
for v in stamplite; do
  MCRT_LIB=$PWD/mcray-tracing_amd/build/libmcrt_hip_$v.so timeout 300 python tools/stamps.py 1024 20 > gpurun_out/r5/stamps_$v.txt 2>&1
  tail -11 gpurun_out/r5/stamps_$v.txt
done
