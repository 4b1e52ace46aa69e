set -x
mkdir -p gpurun_out/prof
(timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log)
tail -15 gpurun_out/pytest_gpu.log
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/prof/counters_list.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/r1_stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof/r1_stats_bench.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/prof/r1_pmc_sq -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/prof/r1_pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d gpurun_out/prof/r1_pmc_fetch -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/prof/r1_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --output-format csv -d gpurun_out/prof/r1_pmc_write -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/prof/r1_pmc_write.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/prof/r1_pmc_sq2 -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/prof/r1_pmc_sq2.log 2>&1
rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum --output-format csv -d gpurun_out/prof/r1_pmc_ta -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/prof/r1_pmc_ta.log 2>&1
find gpurun_out/prof -name "*.csv" | head -50
du -sh gpurun_out/prof
