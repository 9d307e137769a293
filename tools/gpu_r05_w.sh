#!/bin/bash
# round 5, call W: the pipelined K loop of the chain kernel (one workgroup per compute unit: four LDS tile buffers, one barrier per slab,
# the next slab's LDS refill and the loads of the slab after it between the MFMAs): timeline, chain / two-level / pivot tests, phases
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
timeout 600 python tools/chain_trace.py --skip 3 --launches 1 50000 blocked 20000 blocked --skip 0 --launches 1 20000 blocked
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_twolevel.py -q -m gpu -x 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -q -m gpu -x -k "posdef or single_rank or virtual or sharded_model or fit_sizes or lookahead or predict_fp64 or loo" 2>&1 | grep -E "passed|failed"
echo "== blocked one rank: per-step phases"
timeout 600 python bench.py --steps 3 --warmup 1 --mode sharded --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('  N=50000 ms/step %.1f' % j['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in j['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')}); c=j['c2']; print('  c2 ms/step %.2f' % c['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in c['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')})"
echo "== dense"
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('  N=50000 ms/step %.1f frac %.3f; c2 %.2f frac %.3f' % (j['ms_per_step'], j['roofline']['frac'], j['c2']['ms_per_step'], j['c2']['roofline_frac']))"
} > $O/r05_w_chain_pipelined.log 2>&1
cat $O/r05_w_chain_pipelined.log
