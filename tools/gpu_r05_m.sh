#!/bin/bash
# round 5, call M: the chain kernel after its task numbering moved into chain_order.h (a refactoring): its tests, the suites that factor through it, bench
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_twolevel.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_dist.py -q -m gpu -x -k "posdef or single_rank or virtual or sharded_model" 2>&1 | grep -E "passed|failed"
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('N=50000 ms/step %.1f frac %.3f; c2 %.2f frac %.3f' % (j['ms_per_step'], j['roofline']['frac'], j['c2']['ms_per_step'], j['c2']['roofline_frac']))"
} > $O/r05_m_refactor_check.log 2>&1
cat $O/r05_m_refactor_check.log
