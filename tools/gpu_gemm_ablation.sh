#!/bin/bash
# ablation + phase timers of the current update kernel (isolated), both tile shapes, one and two workgroups per CU;
# then the kernel-trace groups of the FITC C5 fit
mkdir -p gpurun_out
{
for w in 2 1; do
  echo "== 128x128 tiles, workgroups per CU: $w"
  GPMI_GEMM_NI=4 GPMI_GEMM_WGS=$w python tools/gemm_phases.py one 0,128,1,2,6,22,512 2>&1 | grep -E "variant|phases|timeline"
done
echo "== 128x128 forced / automatic tile shape, all bench-relevant shapes"
GPMI_GEMM_NI=4 python tools/gemm_phases.py all 0 2>&1 | grep variant
python tools/gemm_phases.py all 0 2>&1 | grep variant
} > gpurun_out/gemm_ablation.log
cat gpurun_out/gemm_ablation.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/gpurun_out/prof_fitc" -- python "$GRAFT_REPO_ROOT/tools/fitc_bench.py" 1000000x4096 > "$GRAFT_REPO_ROOT/gpurun_out/prof_fitc.log" 2>&1
cd "$GRAFT_REPO_ROOT"; DB=$(find gpurun_out/prof_fitc -name "*_results.db" | head -1)
python tools/rocpd_groups.py "$DB" > gpurun_out/fitc_groups.txt; head -14 gpurun_out/fitc_groups.txt
rm -rf gpurun_out/prof_fitc
