#!/bin/bash
# round 6, call Z: L2 hit rate and L2 read requests of the trailing-update class at the bench size (where the 3.6x HBM traffic comes from)
set -u
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/pmc_l2"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$O/tcc" -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$O/tcc.log" 2>&1; echo "tcc exit $?"
timeout 400 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d "$O/tcp" -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$O/tcp.log" 2>&1; echo "tcp exit $?"
cd "$R"; python tools/pmc_l2_update.py "$O" > gpurun_out/r06_update_l2.json; cat gpurun_out/r06_update_l2.json
find "$O" -name "*.csv" -size +20M -delete
