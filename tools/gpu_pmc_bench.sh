#!/bin/bash
# HBM traffic of the bench's dominant kernel, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE
# --pmc passes (with --kernel-trace only) over the bench command itself; the gfx950 correction (FETCH_SIZE x 2 for wide
# coalesced reads) is applied in tools/pmc_summary.py.  Two more passes give the MFMA-busy fraction of the same kernel
# (SQ_VALU_MFMA_BUSY_CYCLES over 1024 SIMDs x the kernel's GRBM_GUI_ACTIVE cycles per XCD).
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/pmc_bench"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" GRBM_GUI_ACTIVE; do
  d=${c%% *}
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/$d" -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$O/$d.log" 2>&1
  echo "pmc $c exit $?"
done
cd "$R"
python tools/pmc_summary.py "$O" > gpurun_out/bench_pmc_hbm.json
cat gpurun_out/bench_pmc_hbm.json
find "$O" -name "*.csv" -size +20M -delete
