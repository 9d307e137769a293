#!/bin/bash
# round 6, call H (the code it measures was REMOVED after this run — slower; the log is profiles/r06_h_*): the fit's backward substitution by super-panels (two launches per stored inverse instead of one per 256 columns) —
# the suites that compare alpha with the oracle, then A/B on the dense bench line (GPMI_BSOLVE_WIDE = 0 | 1).
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_twolevel.py tests/test_gpu_chain.py tests/test_reference_goldens.py "tests/test_gpu_fullsize.py::test_c2_n20000_direct_vs_oracle" "tests/test_gpu_fullsize.py::test_c4_fp32_n20000_d16_vs_fp64_oracle" -m gpu -x -q 2>&1 | tail -6
for bw in 0 1 0 1; do
echo "== dense, GPMI_BSOLVE_WIDE=$bw"
GPMI_BSOLVE_WIDE=$bw timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 > $O/r06_h_bench_dense_bw$bw.json
python -c "import json; j=json.load(open('$O/r06_h_bench_dense_bw$bw.json')); r=j['roofline']; print('  N=50000 ms/step %.1f fits/s %.4f frac %.3f fit %.1f predict %.1f; c2 %.2f fit %.2f predict %.2f frac %.3f' % (j['ms_per_step'], j['value'], r['frac'], j.get('fit_only_ms_per_step',0), j.get('predict_only_ms_per_step',0), j['c2']['ms_per_step'], j['c2']['fit_only_ms_per_step'], j['c2']['predict_only_ms_per_step'], j['c2']['roofline_frac']))"
done
} > $O/r06_h.log 2>&1
cat $O/r06_h.log
