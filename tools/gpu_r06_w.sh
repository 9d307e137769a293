#!/bin/bash
# round 6, call W (the code it measures was REMOVED after this run — slower; log: profiles/r06_w_*): the refined panel step of FITC's whitening as ONE launch per 256 columns (panel.hip rows256r_kernel): the FITC suite, the
# reference's FITC golden, smoke; then A/B (GPMI_ROWS256R = 0 | 1) on C5 (N = 1e6, M = 4096).
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_fitc.py tests/test_reference_goldens.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -1
for rr in 0 1 0 1; do
echo "== C5, GPMI_ROWS256R=$rr"
GPMI_ROWS256R=$rr timeout 300 python tools/fitc_bench.py 1000000x4096 2>&1 | grep -v amdgpu | tail -2
done
} > $O/r06_w.log 2>&1
cat $O/r06_w.log
