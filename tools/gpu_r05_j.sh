#!/bin/bash
# round 5, call J: chain_wait_kernel (the chain's workgroups placed before the update takes the compute units): does the free-slot mode become
# independent of the instrumentation?  bench.py at N = 20 000 / 28 000 / 50 000 with none / all / syrk events; chain tests; blocked one rank
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_twolevel.py -q -m gpu -x 2>&1 | tail -3
for N in 20000 28000 50000; do
for H in none all syrk; do
  echo "== bench.py --n $N, events in the timed region: $H"
  GPMI_BENCH_PROFILE=$H timeout 300 python bench.py --n $N --steps 4 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('  ms/step %.2f fit %.2f predict %.2f  roofline %.3f launches %d' % (j['ms_per_step'], j['fit_only_ms_per_step'], j['predict_only_ms_per_step'], j['roofline']['frac'], j['roofline']['launches']))"
done; done
echo "== blocked one rank"
timeout 600 python bench.py --steps 3 --warmup 1 --mode sharded --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('  N=50000 ms/step %.1f' % j['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in j['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')}); c=j['c2']; print('  c2 ms/step %.2f' % c['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in c['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')})"
} > $O/r05_j_chain_wait.log 2>&1
cat $O/r05_j_chain_wait.log
