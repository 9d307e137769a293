#!/bin/bash
# round 6, call D: the blocked handle on one rank with the longer chain_wait (full-grid update), then the GPU suite as the driver runs it,
# full-size tests first (tests/conftest.py), with durations.
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
echo "== blocked one rank: per-step phases"
timeout 300 python bench.py --steps 3 --warmup 1 --mode sharded --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 > $O/r06_d_bench_blocked.json
python -c "
import json; j=json.load(open('$O/r06_d_bench_blocked.json')); print('  N=50000 ms/step %.1f' % j['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in j['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')}); c=j['c2']; print('  c2 ms/step %.2f' % c['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in c['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')})"
echo "== dense"
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 > $O/r06_d_bench_dense.json
python -c "import json; j=json.load(open('$O/r06_d_bench_dense.json')); r=j['roofline']; print('  N=50000 ms/step %.1f fits/s %.4f frac %.3f; c2 %.2f frac %.3f' % (j['ms_per_step'], j['value'], r['frac'], j['c2']['ms_per_step'], j['c2']['roofline_frac']))"
echo "== GPU suite"
( time python -m pytest tests -m gpu -x -q --durations=25 ) > $O/r06_d_gpu_tests.log 2>&1
tail -45 $O/r06_d_gpu_tests.log
cat $O/test_laps.log
} > $O/r06_d.log 2>&1
cat $O/r06_d.log
