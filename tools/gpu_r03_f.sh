#!/bin/bash
# round 3, call F: stream order prio / own / masked / masked; per-step hybrid in the dense path; blocked critical path
mkdir -p gpurun_out; O=gpurun_out
run() { env "$@" timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary c2 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$*', 'n50000 ms %.1f frac %.3f | c2 ms %.2f frac %.3f' % (j['ms_per_step'], j['roofline']['frac'], j['c2']['ms_per_step'], j['c2']['roofline_frac']))"; }
{ run GPMI_X=default; run GPMI_CUMASK=0; } 2>&1 | tee $O/f_streams.log
echo "== blocked code path on one rank"
timeout 600 python bench.py --mode sharded --steps 5 --warmup 2 --no-cpu-baseline --secondary c2 2> $O/f_sharded.err | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('sharded world 1: n50000 ms %.1f frac %.3f mll %.6f | c2 ms %.2f' % (j['ms_per_step'], j['roofline']['frac'], j['config']['mll'], j['c2']['ms_per_step']))" 2>&1 | tee $O/f_sharded.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/prof_f" -- python "$GRAFT_REPO_ROOT/bench.py" --mode sharded --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$GRAFT_REPO_ROOT/$O/prof_f.log" 2>&1
cd "$GRAFT_REPO_ROOT"
DB=$(find $O/prof_f -name "*_results.db" | head -1); python tools/rocpd_mainstream.py "$DB" > $O/f_sharded_critical_path.txt 2>&1; head -30 $O/f_sharded_critical_path.txt | cut -c1-150; tail -4 $O/f_sharded_critical_path.txt | cut -c1-300
cp "$DB" $O/f_sharded.db 2>/dev/null; ls -la $O/f_sharded.db
rm -rf $O/prof_f
