#!/bin/bash
# round 3, call H: where the blocked driver's world-1 time goes at N = 50000 (2048- and 1024-row blocks)
mkdir -p gpurun_out; O=gpurun_out
cd /tmp && export TMPDIR=/tmp
for wd in 2048 1024; do
  GPMI_DIST_WD=$wd timeout 300 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/prof_h$wd" -- python "$GRAFT_REPO_ROOT/bench.py" --mode sharded --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$GRAFT_REPO_ROOT/$O/prof_h$wd.log" 2>&1
  DB=$(find $GRAFT_REPO_ROOT/$O/prof_h$wd -name "*_results.db" | head -1)
  echo "=== WD = $wd: $(grep -o '"ms_per_step": [0-9.]*' $GRAFT_REPO_ROOT/$O/prof_h$wd.log | head -1)"
  python $GRAFT_REPO_ROOT/tools/rocpd_blocked.py "$DB" 2>&1 | tee $GRAFT_REPO_ROOT/$O/h_blocked_wd$wd.txt
  rm -rf $GRAFT_REPO_ROOT/$O/prof_h$wd
done
