#!/bin/bash
# round 4, call F: the whole -m gpu suite (no -x: every failure is listed); FITC C5 with the tall products on update256_kernel; predict_f's
# updates through the rectangular 256 x 128 form (GPMI_UPDATE256_RECT=1024) against the default
mkdir -p gpurun_out; O=gpurun_out
timeout 2700 python -m pytest tests -q -m gpu -s > $O/r04_f_tests.log 2>&1; grep -v amdgpu $O/r04_f_tests.log | grep -E "passed|failed|^FAILED|^ERROR|injected-latency|C4 N" | cut -c1-600 | tail -15
echo "== C5 (FITC N = 1e6, M = 4096)"
for u in 1 0; do
GPMI_UPDATE256=$u timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --secondary c5 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline()); c = j['c5']
print('update256=$u c5: fit %.3f s (%.3f of peak) predict %.2f ms grad %.3f s parity' % (c['update_mll_s'], c['update_mll_frac_of_fp64_matrix_peak'], c['predict_f_1024_ms'], c['update_dmll_s']), c['parity']['rel_err'], c['parity']['ok'], '| n50000 ms %.1f' % j['ms_per_step'])"
done 2>&1 | tee $O/r04_f_c5.log
echo "== predict through the rectangular 256 x 128 form"
for r in 8192 1024; do
GPMI_UPDATE256_RECT=$r timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary c2 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('rect_min_m=$r: n50000 ms %.1f predict %.2f | c2 ms %.2f predict %.2f' % (j['ms_per_step'], j['predict_only_ms_per_step'], j['c2']['ms_per_step'], j['c2']['predict_only_ms_per_step']))"
done 2>&1 | tee $O/r04_f_predict_rect.log
