#!/bin/bash
# PMC passes for one kernel (PMC_KERNEL, default k_trace_lane<false>): separate rocprofv3 runs per counter group, as the pool requires
export MCRT_TUNING=1      # (MCRT_WIDE_FROM below is a tuning knob)
out=gpurun_out/pmc_$1; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp      # (a fresh directory: rocprofv3 names its files by process id, an older run's would be summarised too)
B="python3 bench.py --steps 128 --warmup 128 --no-cpu-baseline --no-latency-leg --no-pmc ${BENCH_ARGS}"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 128 --warmup 128 --no-cpu-baseline --no-latency-leg --no-pmc ${BENCH_ARGS} > $out/stats.log 2>&1
timeout -s KILL 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $out/sq1 -- $B > $out/sq1.log 2>&1
timeout -s KILL 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $out/sq2 -- $B > $out/sq2.log 2>&1
timeout -s KILL 400 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS SQ_INSTS_FLAT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $out/sq3 -- $B > $out/sq3.log 2>&1
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $out/fetch -- $B > $out/fetch.log 2>&1
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/write -- $B > $out/write.log 2>&1
timeout -s KILL 400 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $out/tcp -- $B > $out/tcp.log 2>&1
MCRT_WIDE_FROM=4294967295 timeout -s KILL 400 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $out/tcp_narrow -- $B > $out/tcp_narrow.log 2>&1   # the walk's accesses without the five-wavefront form's spill traffic (counters named ...@narrow)
timeout -s KILL 400 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $out/grbm -- $B > $out/grbm.log 2>&1
python3 tools/pmc_summary.py $out
