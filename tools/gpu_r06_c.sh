#!/bin/bash
# round 6, call C: the full driver-like line (all secondaries) at the tree with potf2_wg + the update's full grid, the blocked handle on one
# rank (its update now also takes every compute unit), then the three long tests with their stage clocks (where the suite's minutes go).
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
echo "== bench.py (defaults: what the driver runs)"
timeout 900 python bench.py 2>$O/r06_c_bench.err | tail -1 > $O/r06_c_bench.json
python - <<'PY'
import json
j = json.load(open('gpurun_out/r06_c_bench.json')); r = j['roofline']
print('  N=50000 ms/step %.1f fits/s %.4f frac %.3f peak_measured %.1f frac_of_measured %.3f fit %.1f predict %.1f' % (j['ms_per_step'], j['value'], r['frac'], r.get('peak_measured', 0), r.get('frac_of_measured', 0), j.get('fit_only_ms_per_step', 0), j.get('predict_only_ms_per_step', 0)))
for k in ('c2', 'c3', 'c4_single_gpu', 'grad', 'c5'):
    v = j.get(k)
    if isinstance(v, dict):
        print('  ', k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if isinstance(b, (int, float))})
print('   parity', j.get('parity', {}).get('ok'), 'cpu', j.get('cpu_baseline', {}).get('value'))
PY
echo "== blocked one rank: per-step phases"
timeout 300 python bench.py --steps 3 --warmup 1 --mode sharded --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 > $O/r06_c_bench_blocked.json
python -c "
import json; j=json.load(open('$O/r06_c_bench_blocked.json')); print('  N=50000 ms/step %.1f' % j['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in j['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')}); c=j['c2']; print('  c2 ms/step %.2f' % c['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in c['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')})"
echo "== the three long tests, stage clocks"
rm -f $O/test_laps.log
timeout 1100 python -m pytest "tests/test_gpu_fullsize.py::test_c3_n50000_composite_direct_vs_oracle" "tests/test_gpu_fullsize.py::test_f3_packed_n220000_fp64_on_one_device_block_diagonal" "tests/test_gpu_fullsize.py::test_c4_fp32_n200000_d16_properties_at_full_size" -m gpu -x -q -s --durations=5 2>&1 | grep -v "^$" | tail -25
cat $O/test_laps.log
} > $O/r06_c.log 2>&1
cat $O/r06_c.log
