#!/bin/bash
# round 5, call C: with the fast chain, where do the super-panel thresholds, the whole-CU / free-slot switch, the look-ahead minimum and the
# fused tail want to be?  (dense C2 and N = 50 000; the blocked handle with 1024 / 2048-row blocks on one rank)
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
echo "== tests of the fused tail (every chain test + fit sizes)"
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_parity.py tests/test_gpu_twolevel.py -q -m gpu -x -k "chain or fit_sizes or width or lookahead or posdef or predict_fp64 or loo" 2>&1 | tail -5
echo "== dense N = 20000"
timeout 900 python tools/knob_sweep.py 20000 dense 2>&1 | grep "^N="
echo "== dense N = 50000"
timeout 900 python tools/knob_sweep.py 50000 dense 2>&1 | grep "^N="
echo "== blocked, one rank"
timeout 600 python tools/knob_sweep.py 50000 blocked 2>&1 | grep "^N="
timeout 600 python tools/knob_sweep.py 20000 blocked 2>&1 | grep "^N="
} > $O/r05_c_sweeps.log 2>&1
cat $O/r05_c_sweeps.log
