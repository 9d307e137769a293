#!/bin/bash
# round 3, call B: the blocked driver below the ABI on the device, hybrid masked / free-slot look-ahead, world-1 overhead
mkdir -p gpurun_out; O=gpurun_out
echo "== blocked / sharded GPU tests"
timeout 1500 python -m pytest tests/test_gpu_dist.py -q -m gpu -x --durations=6 > $O/b_tests_dist.log 2>&1; grep -v amdgpu $O/b_tests_dist.log | tail -25
echo "== smoke"
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -3
echo "== dense path: hybrid look-ahead (default) and thresholds"
for cfg in "GPMI_CUMASK=0" "GPMI_X=default" "GPMI_CUMASK_BELOW=14336" "GPMI_CUMASK_BELOW=28672"; do
  env $cfg timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary c2 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$cfg', 'n50000 ms %.1f frac %.3f | c2 ms %.2f frac %.3f' % (j['ms_per_step'], j['roofline']['frac'], j['c2']['ms_per_step'], j['c2']['roofline_frac']))"
done 2>&1 | tee $O/b_ab.log
echo "== blocked code path on one rank (world-1 overhead): N=50000 + c2 + c4 size"
timeout 600 python bench.py --mode sharded --steps 5 --warmup 2 --no-cpu-baseline --secondary c2,c4 2> $O/b_sharded.err | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('sharded world 1: n50000 ms %.1f frac %.3f mll %.6f | c2 ms %.2f | c4 %s' % (j['ms_per_step'], j['roofline']['frac'], j['config']['mll'], j['c2']['ms_per_step'], json.dumps(j.get('c4_single_gpu'))))
print(json.dumps(j['stage_ms_per_step']))" 2>&1 | tee $O/b_sharded.log
tail -3 $O/b_sharded.err | grep -v amdgpu
echo "== dense c4 for comparison"
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --secondary c4 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline()); print('dense: n50000 ms %.1f mll %.6f | c4 %s' % (j['ms_per_step'], j['config']['mll'], json.dumps(j.get('c4_single_gpu'))))" 2>&1 | tee -a $O/b_sharded.log
echo "== packed N = 220000 through gpmi_fit"
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -s -k "f3_packed" > $O/b_f3.log 2>&1; grep -v amdgpu $O/b_f3.log | tail -4
echo "== blocked gradient timing at N = 50000 (one rank, plain rows and packed)"
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu | tee $O/b_grad.log
import math, sys, time, os
sys.path.insert(0, "gaussianprocesses.jl_amd"); sys.path.insert(0, ".")
import numpy as np
import gpmi355x as g
from gpmi355x import dist as gd
rng = np.random.default_rng(20240501)
n, d = 50000, 8
x = rng.uniform(size=(d, n)); y = np.sin(2 * np.pi * x).sum(axis=0) / d + 0.1 * rng.standard_normal(n)
ll = [math.log(0.5) + 0.05 * k for k in range(d)]
dense = g.GP(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1)); dense.update_dmll()
t0 = time.perf_counter(); dense.update_dmll(); td = time.perf_counter() - t0
ref = dense.dmll.copy(); del dense
for kw in (dict(), dict(stripe_blocks=8)):
    gp = gd.ShardedGPE(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1), block=1024, **kw)
    t0 = time.perf_counter(); gp.update_mll(); tf = time.perf_counter() - t0
    gp.update_dmll()
    t0 = time.perf_counter(); gp.update_dmll(); tg = time.perf_counter() - t0
    print(f"blocked {kw}: update_mll {tf*1e3:.1f} ms, update_dmll {tg:.3f} s (dense gpmi_grad {td:.3f} s), max rel diff vs dense {np.abs(gp.dmll - ref).max() / np.abs(ref).max():.2e}, factor {gp.cK.factor_bytes/1e9:.1f} GB")
    del gp
PY
echo "== critical path of the blocked fit at N = 50000"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/prof_b" -- python "$GRAFT_REPO_ROOT/bench.py" --mode sharded --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$GRAFT_REPO_ROOT/$O/prof_b.log" 2>&1
cd "$GRAFT_REPO_ROOT"
DB=$(find $O/prof_b -name "*_results.db" | head -1); python tools/rocpd_mainstream.py "$DB" > $O/b_sharded_critical_path.txt; head -14 $O/b_sharded_critical_path.txt; tail -3 $O/b_sharded_critical_path.txt
rm -rf $O/prof_b
