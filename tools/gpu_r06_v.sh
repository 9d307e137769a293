#!/bin/bash
# round 6, call V: where one FITC fit's 0.59 s go — kernel trace of two fits at N = 1e6, M = 4096
mkdir -p gpurun_out; O=gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_fitc" -- python "$GRAFT_REPO_ROOT/tools/fitc_fit_only.py" > "$GRAFT_REPO_ROOT/$O/prof_fitc.log" 2>&1
cd "$GRAFT_REPO_ROOT"; tail -2 $O/prof_fitc.log
DB=$(find $O/prof_fitc -name "*_results.db" | head -1)
python tools/rocpd_stats.py "$DB" > $O/r06_v_fitc_fit_kernel_stats.csv
cut -c1-150 $O/r06_v_fitc_fit_kernel_stats.csv | head -30
rm -rf $O/prof_fitc
