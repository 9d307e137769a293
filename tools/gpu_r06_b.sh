#!/bin/bash
# round 6, call B: the workgroup-wide potf2 (potf2.h potf2_wg) inside the chain kernel — correctness first (chain / two-level / parity suites),
# then the in-kernel timeline (tools/chain_trace.py), then the bench lines with the update's grid covering every compute unit (A/B).
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
echo "== chain / two-level / parity suites"
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_twolevel.py tests/test_gpu_parity.py tests/test_reference_goldens.py -m gpu -x -q --durations=15 2>&1 | tail -30
echo "== chain timeline (trace build)"
timeout 600 python tools/chain_trace.py --skip 3 --launches 1 20000 blocked 50000 blocked --skip 0 --launches 1 20000 blocked
for fg in 0 1; do
echo "== dense, GPMI_UPDATE_FULL_GRID=$fg"
GPMI_UPDATE_FULL_GRID=$fg timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 > $O/r06_b_bench_dense_fg$fg.json
python -c "import json; j=json.load(open('$O/r06_b_bench_dense_fg$fg.json')); r=j['roofline']; print('  N=50000 ms/step %.1f fits/s %.4f frac %.3f peak_measured %s frac_of_measured %s; c2 %.2f frac %.3f' % (j['ms_per_step'], j['value'], r['frac'], r.get('peak_measured'), r.get('frac_of_measured'), j['c2']['ms_per_step'], j['c2']['roofline_frac']))"
done
echo "== blocked one rank: per-step phases"
timeout 300 python bench.py --steps 3 --warmup 1 --mode sharded --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 > $O/r06_b_bench_blocked.json
python -c "
import json; j=json.load(open('$O/r06_b_bench_blocked.json')); print('  N=50000 ms/step %.1f' % j['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in j['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')}); c=j['c2']; print('  c2 ms/step %.2f' % c['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in c['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')})"
} > $O/r06_b.log 2>&1
cat $O/r06_b.log
