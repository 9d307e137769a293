#!/bin/bash
# round 5, final: the records (tools/gpu_r05_records.sh) and the whole GPU suite as the driver runs it
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
bash tools/gpu_r05_records.sh > $O/r05_records.log 2>&1
tail -30 $O/r05_records.log | cut -c1-600
cd "$GRAFT_REPO_ROOT"
( time timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -12 ) > $O/r05_gpu_tests.log 2>&1
cat $O/r05_gpu_tests.log
