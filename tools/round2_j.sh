#!/bin/bash
# gated: verify the restored k_march, then the full profile set
mkdir -p gpurun_out/r2j; export TMPDIR=/tmp
O=gpurun_out/r2j
( time timeout 120 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1 || { echo "GATE smoke failed"; tail -5 $O/smoke.log; exit 1; }
( time timeout 600 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log | cut -c1-200
grep -q "rc=0" $O/pytest.log || { grep -E "^E  |^FAILED" $O/pytest.log | head -10 | cut -c1-300; exit 1; }
bash tools/profile_round.sh r2
