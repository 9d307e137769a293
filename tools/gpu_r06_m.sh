#!/bin/bash
# round 6, call M: more chain workgroups beside a FULL-GRID update (the chain is resident before the update starts, its units go back to the
# update when it exits): GPMI_CHAIN_BESIDE_WGS = 8 | 16 | 32 on the dense line, C2 and the blocked handle on one rank.
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
for w in 8 16 32 8 16; do
echo "== dense, GPMI_CHAIN_BESIDE_WGS=$w"
GPMI_CHAIN_BESIDE_WGS=$w timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 > $O/r06_m_bench_dense_w$w.json
python -c "import json; j=json.load(open('$O/r06_m_bench_dense_w$w.json')); r=j['roofline']; print('  N=50000 ms/step %.1f fits/s %.4f frac %.3f fit %.1f; c2 %.2f fit %.2f frac %.3f' % (j['ms_per_step'], j['value'], r['frac'], j.get('fit_only_ms_per_step',0), j['c2']['ms_per_step'], j['c2']['fit_only_ms_per_step'], j['c2']['roofline_frac']))"
done
for w in 8 16 32; do
echo "== blocked one rank, GPMI_CHAIN_BESIDE_WGS=$w"
GPMI_CHAIN_BESIDE_WGS=$w timeout 300 python bench.py --steps 3 --warmup 1 --mode sharded --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 > $O/r06_m_bench_blocked_w$w.json
python -c "
import json; j=json.load(open('$O/r06_m_bench_blocked_w$w.json')); print('  N=50000 ms/step %.1f' % j['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in j['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')}); c=j['c2']; print('  c2 ms/step %.2f' % c['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in c['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')})"
done
echo "== chain tests with 16"
GPMI_CHAIN_BESIDE_WGS=16 timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_twolevel.py -m gpu -x -q 2>&1 | tail -3
} > $O/r06_m.log 2>&1
cat $O/r06_m.log
