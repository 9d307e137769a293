#!/bin/bash
# round 3, call E: all streams created with the context in a fixed order; blocked path on one rank with 512-row default blocks below 32768
mkdir -p gpurun_out; O=gpurun_out
run() { env "$@" timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary c2 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$*', 'n50000 ms %.1f frac %.3f | c2 ms %.2f frac %.3f' % (j['ms_per_step'], j['roofline']['frac'], j['c2']['ms_per_step'], j['c2']['roofline_frac']))"; }
{ run GPMI_X=default; run GPMI_CUMASK=0; } 2>&1 | tee $O/e_streams.log
echo "== blocked code path on one rank"
timeout 600 python bench.py --mode sharded --steps 5 --warmup 2 --no-cpu-baseline --secondary c2,c4 2> $O/e_sharded.err | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('sharded world 1: n50000 ms %.1f frac %.3f mll %.6f | c2 ms %.2f | c4 %s' % (j['ms_per_step'], j['roofline']['frac'], j['config']['mll'], j['c2']['ms_per_step'], json.dumps(j.get('c4_single_gpu'))))
print(json.dumps(j['stage_ms_per_step']))" 2>&1 | tee $O/e_sharded.log
timeout 600 python -m pytest tests/test_gpu_dist.py -q -m gpu -x > $O/e_tests.log 2>&1; grep -v amdgpu $O/e_tests.log | tail -4
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/prof_e" -- python "$GRAFT_REPO_ROOT/bench.py" --mode sharded --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$GRAFT_REPO_ROOT/$O/prof_e.log" 2>&1
cd "$GRAFT_REPO_ROOT"
DB=$(find $O/prof_e -name "*_results.db" | head -1); python tools/rocpd_stats.py "$DB" > $O/e_sharded_kernel_stats.csv; head -14 $O/e_sharded_kernel_stats.csv | cut -c1-160
rm -rf $O/prof_e
