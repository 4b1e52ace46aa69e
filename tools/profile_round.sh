#!/bin/bash
# everything profiles/<round>/ is packaged from (tools/package_profiles.py): the bench line (with its live PMC passes, CPU baseline
# and parity check), the driver's own command, kernel statistics overlapped and standalone, PMC passes, timelines, the BASELINE configs
tag=${1:-round}; export TMPDIR=/tmp; export MCRT_TUNING=1      # (MCRT_NO_OVERLAP below is a tuning knob)
O=gpurun_out/profile_$tag; mkdir -p $O
( time timeout 600 python bench.py ) > $O/bench.json 2> $O/bench.err
( time timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
bash tools/pmc.sh $tag > $O/pmc_trace.txt 2>&1
MCRT_NO_OVERLAP=1 bash tools/kstats.sh ${tag}_alone > $O/kernels_standalone.txt 2>&1
bash tools/timeline.sh $tag > $O/frame_timeline.txt 2>&1
BENCH_ARGS="--frames-in-flight 1" bash tools/timeline.sh ${tag}_f1 > $O/frame_timeline_one_frame.txt 2>&1
bash tools/configs.sh > $O/configs.txt 2>&1
( time timeout 900 python bench.py --workload random16m ) > $O/bench_random16m.json 2> $O/bench_random16m.err
timeout 600 python tools/group_bench.py 0,0 32 8 > $O/group_bench.json 2> $O/group_bench.err
tail -3 $O/bench.err; cat $O/configs.txt; head -5 $O/kernels_standalone.txt
