#!/bin/bash
# round 4, call J: the whole -m gpu suite at the final tree, then the records (tools/gpu_r04_records.sh)
mkdir -p gpurun_out; O=gpurun_out
timeout 2700 python -m pytest tests -q -m gpu -s --durations=8 > $O/r04_j_tests.log 2>&1; grep -v amdgpu $O/r04_j_tests.log | grep -E "passed|failed|^FAILED|^ERROR|injected-latency|C4 N|s call" | cut -c1-700 | tail -16
bash tools/gpu_r04_records.sh
