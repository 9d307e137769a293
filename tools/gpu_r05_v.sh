#!/bin/bash
# round 5, call V: potf2 with the off-diagonal inverse blocks of one distance issued together and 1/L_jj as a global store (potf2.h): phase marks, timeline, chain / two-level / pivot tests
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
timeout 600 python tools/chain_trace.py --skip 3 --launches 1 50000 blocked 20000 blocked --skip 0 --launches 1 20000 blocked
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_twolevel.py -q -m gpu -x 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -q -m gpu -x -k "posdef or single_rank or virtual or sharded_model or fit_sizes or lookahead or predict_fp64 or loo" 2>&1 | grep -E "passed|failed"
} > $O/r05_v_potf2_offdiag.log 2>&1
cat $O/r05_v_potf2_offdiag.log
