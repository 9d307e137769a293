#!/bin/bash
# round 6, call G: the tall panel solves X LW' in 256 x 128 tiles (update256_kernel's KEND instantiation) — the new test and the suites that
# factor through it, then A/B on the dense bench line (GPMI_UPDATE256_KEND = 0 | 1).
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_twolevel.py tests/test_gpu_parity.py tests/test_gpu_chain.py "tests/test_gpu_fullsize.py::test_c2_n20000_direct_vs_oracle" -m gpu -x -q 2>&1 | tail -6
for kd in 0 1 0 1; do
echo "== dense, GPMI_UPDATE256_KEND=$kd"
GPMI_UPDATE256_KEND=$kd timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 > $O/r06_g_bench_dense_kend$kd.json
python -c "import json; j=json.load(open('$O/r06_g_bench_dense_kend$kd.json')); r=j['roofline']; print('  N=50000 ms/step %.1f fits/s %.4f frac %.3f fit %.1f predict %.1f; c2 %.2f fit %.2f predict %.2f frac %.3f' % (j['ms_per_step'], j['value'], r['frac'], j.get('fit_only_ms_per_step',0), j.get('predict_only_ms_per_step',0), j['c2']['ms_per_step'], j['c2']['fit_only_ms_per_step'], j['c2']['predict_only_ms_per_step'], j['c2']['roofline_frac']))"
done
} > $O/r06_g.log 2>&1
cat $O/r06_g.log
