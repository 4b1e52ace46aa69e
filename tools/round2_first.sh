#!/bin/bash
# first GPU call of round 2: full GPU test suite, VALU calibration (+ its PMC cross-check), the default bench line
mkdir -p gpurun_out/r2a; export TMPDIR=/tmp
( time python -m pytest tests -m gpu -x -q --durations=20 ) > gpurun_out/r2a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
build/valu_roof > gpurun_out/r2a/valu_roof.json 2> gpurun_out/r2a/valu_roof.err
( cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2a/valu_pmc -- $GRAFT_REPO_ROOT/build/valu_roof --quick > $GRAFT_REPO_ROOT/gpurun_out/r2a/valu_pmc.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2a/valu_pmc.err )
( time python bench.py ) > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
( time python bench.py --steps 20 --warmup 5 --no-pmc ) > gpurun_out/r2a/bench_k20.json 2> gpurun_out/r2a/bench_k20.err
tail -5 gpurun_out/r2a/pytest.log; cat gpurun_out/r2a/bench.json | cut -c1-1500
