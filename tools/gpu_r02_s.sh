#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_twolevel.py -q -m gpu -x > gpurun_out/s_tests.log 2>&1; tail -2 gpurun_out/s_tests.log
GPMI_PHASE_LOCK=1024 timeout 600 python -m pytest tests/test_gpu_twolevel.py -q -m gpu -x > gpurun_out/s_tests2.log 2>&1; tail -2 gpurun_out/s_tests2.log
for pl in 0 1024 2048 512; do
  GPMI_PHASE_LOCK=$pl timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read()); s=j['stage_ms_per_step']
print('phase_lock_min_k=$pl', 'ms', round(j['ms_per_step'],2), 'upd TF', round(j['roofline']['achieved'],1), 'upd ms', round(s['chol_trailing_update'],1), 'mll', repr(j['config']['mll']))"
done 2>&1 | tee gpurun_out/phase_lock.log
cd /tmp && export TMPDIR=/tmp
for pl in 0 1024; do
for c in FETCH_SIZE WRITE_SIZE; do
  GPMI_PHASE_LOCK=$pl timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pl_$pl/$c" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > /dev/null 2>&1
done
python "$GRAFT_REPO_ROOT/tools/pmc_summary.py" "$GRAFT_REPO_ROOT/gpurun_out/pl_$pl" | tee -a "$GRAFT_REPO_ROOT/gpurun_out/phase_lock.log"
rm -rf "$GRAFT_REPO_ROOT/gpurun_out/pl_$pl"
done
