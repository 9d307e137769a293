#!/bin/bash
# round 6, call E: GEMM_KEND_COL launches (panel solves X LW', predict's V = R LW') walking every strip from its longest-K column — A/B on the
# dense bench line, C2 and the blocked handle; then the parity suites that go through those launches.
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
for hf in 0 1 0 1; do
echo "== dense, GPMI_KEND_HEAVY_FIRST=$hf"
GPMI_KEND_HEAVY_FIRST=$hf timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 > $O/r06_e_bench_dense_hf$hf.json
python -c "import json; j=json.load(open('$O/r06_e_bench_dense_hf$hf.json')); r=j['roofline']; print('  N=50000 ms/step %.1f fits/s %.4f frac %.3f fit %.1f predict %.1f; c2 %.2f fit %.2f predict %.2f frac %.3f' % (j['ms_per_step'], j['value'], r['frac'], j.get('fit_only_ms_per_step',0), j.get('predict_only_ms_per_step',0), j['c2']['ms_per_step'], j['c2']['fit_only_ms_per_step'], j['c2']['predict_only_ms_per_step'], j['c2']['roofline_frac']))"
done
echo "== blocked one rank, heavy first"
timeout 300 python bench.py --steps 3 --warmup 1 --mode sharded --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 > $O/r06_e_bench_blocked.json
python -c "
import json; j=json.load(open('$O/r06_e_bench_blocked.json')); print('  N=50000 ms/step %.1f' % j['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in j['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')}); c=j['c2']; print('  c2 ms/step %.2f' % c['ms_per_step'], {n: round(v['ms_per_block_step'], 3) for n, v in c['per_step_ms'].items() if isinstance(v, dict) and v.get('ms_per_block_step')})"
echo "== parity suites through the changed launches"
timeout 900 python -m pytest tests/test_gpu_twolevel.py tests/test_gpu_parity.py tests/test_gpu_chain.py tests/test_reference_goldens.py -m gpu -x -q 2>&1 | tail -4
} > $O/r06_e.log 2>&1
cat $O/r06_e.log
