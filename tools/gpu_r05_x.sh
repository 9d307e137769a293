#!/bin/bash
# round 5, call X: the tree with the chain kernel's new task order / slab product / potf2 change — bench lines first, then as much of the
# GPU suite as the remaining minutes allow (files that factor through the changed kernels first; -x; durations listed)
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
echo "== dense"
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 > $O/r05_x_bench_dense.json
python -c "import json; j=json.load(open('$O/r05_x_bench_dense.json')); print('  N=50000 ms/step %.1f fits/s %.4f frac %.3f; c2 %.2f frac %.3f' % (j['ms_per_step'], j['value'], j['roofline']['frac'], j['c2']['ms_per_step'], j['c2']['roofline_frac']))"
echo "== blocked one rank: per-step phases"
timeout 300 python bench.py --steps 3 --warmup 1 --mode sharded --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 > $O/r05_x_bench_blocked.json
python -c "
import json; j=json.load(open('$O/r05_x_bench_blocked.json')); print('  N=50000 ms/step %.1f' % j['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in j['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')}); c=j['c2']; print('  c2 ms/step %.2f' % c['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in c['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')})"
echo "== GPU suite (ordered; stops at the first failure or at the time limit)"
timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_twolevel.py tests/test_gpu_parity.py tests/test_reference_goldens.py tests/test_gpu_fitc.py tests/test_gpu_dist.py tests/test_gpu_fullsize.py -m gpu -x -q --durations=20 2>&1 | tail -40
} > $O/r05_x_final.log 2>&1
cat $O/r05_x_final.log
