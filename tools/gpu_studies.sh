#!/bin/bash
# raw output of the studies DESIGN.md quotes -> gpurun_out/studies/*.log (copied to profiles/ by hand)
O=gpurun_out/studies; mkdir -p $O
{ for w in 2 1; do echo "== GPMI_GEMM_WGS=$w (128x128 tiles forced)"; GPMI_GEMM_NI=4 GPMI_GEMM_WGS=$w timeout 120 python tools/gemm_phases.py one 0,128 2>&1 | grep -v amdgpu; done
  echo "== ablation, two workgroups per CU then one (variants: 0 full, 1 no C read, 2 no epilogue, 6 +no DMA, 22 +no LDS fragment reads, 30 +no rotations)"
  for w in 2 1; do GPMI_GEMM_NI=4 GPMI_GEMM_WGS=$w timeout 120 python tools/gemm_phases.py one 0,1,2,6,22,30 2>&1 | grep variant; done
  echo "== tile shapes: GPMI_GEMM_NI=4 then 2"
  for ni in 4 2; do GPMI_GEMM_NI=$ni timeout 120 python tools/gemm_phases.py all 0 2>&1 | grep variant; done; } > $O/gemm_phases.log 2>&1
{ timeout 60 tools/bin/slot_probe; timeout 60 tools/bin/cumask_probe 8; timeout 60 tools/bin/cumask_probe 32; } > $O/coresidency_cumask_probe.log 2>&1
{ for r in 0 1; do GPMI_REFINE=$r timeout 120 python tools/illcond_check.py 2>&1 | grep -v amdgpu; done
  timeout 250 python tools/small_noise_check.py 2>&1 | grep -v amdgpu
  timeout 250 python tools/fitc_probe.py 2>&1 | grep -v amdgpu; } > $O/conditioning_studies.log 2>&1
for f in $O/*.log; do tail -n 2 $f; done
