#!/bin/bash
# round 6, call P: whole-CU threshold 12288 in the dense path (tests that straddle it), and where the ONE-rank blocked handle should switch
# (GPMI_BLOCKED_WHOLE_BELOW = 20480 (default) | 12288 | 0) at N = 20 000 and 50 000.
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_twolevel.py tests/test_gpu_chain.py "tests/test_gpu_fullsize.py::test_c2_n20000_direct_vs_oracle" -m gpu -x -q 2>&1 | tail -4
for wb in 20480 12288 0 20480 12288; do
echo "== blocked one rank, GPMI_BLOCKED_WHOLE_BELOW=$wb"
GPMI_BLOCKED_WHOLE_BELOW=$wb timeout 300 python bench.py --steps 3 --warmup 1 --mode sharded --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 > $O/r06_p_bench_blocked_wb$wb.json
python -c "
import json; j=json.load(open('$O/r06_p_bench_blocked_wb$wb.json')); print('  N=50000 ms/step %.1f' % j['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in j['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')}); c=j['c2']; print('  c2 ms/step %.2f' % c['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in c['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')})"
done
echo "== dense"
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 > $O/r06_p_bench_dense.json
python -c "import json; j=json.load(open('$O/r06_p_bench_dense.json')); r=j['roofline']; print('  N=50000 ms/step %.1f fits/s %.4f frac %.3f; c2 %.2f frac %.3f' % (j['ms_per_step'], j['value'], r['frac'], j['c2']['ms_per_step'], j['c2']['roofline_frac']))"
} > $O/r06_p.log 2>&1
cat $O/r06_p.log
