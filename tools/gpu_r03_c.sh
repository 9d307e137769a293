#!/bin/bash
# round 3, call C: virtual-rank failure of call B (suspect: 2-D blit of a few very long rows), one stream set at a time
mkdir -p gpurun_out; O=gpurun_out
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu | tee $O/c_debug.log
import math, sys, threading, os
sys.path.insert(0, "gaussianprocesses.jl_amd"); sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import gpmi355x as g
from gpmi355x import dist as gd
from dist_helpers import LocalThreadComm
from oracle import gp_oracle as G
SPEC = ("sum", ("se_ard", [-0.5, -0.3, -0.6, -0.2], 0.2), ("mat52_iso", -0.4, -0.5))
rng = np.random.default_rng(3)
for world, n, block in [(2, 1000, None), (3, 1793, None)]:
    x = rng.uniform(size=(4, n)); y = np.sin(3 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
    ln = math.log(0.1)
    ref = G.update_mll(SPEC, x, y, ln, ("const", 0.1))
    shared = LocalThreadComm.Shared(world)
    def run(rank):
        try:
            ctx = g.Context(0)
            comm = LocalThreadComm(shared, rank)
            gp = gd.ShardedGPE(x, y, g.MeanConst(0.1), g.from_spec(SPEC), ln, comm=comm, ctx=ctx, block=block)
            dg = gp.cK.factor_diag(); WD = gp.WD
            e = np.abs(dg - np.diag(ref["U"])) / np.diag(ref["U"])
            ea = np.abs(gp.alpha - ref["alpha"])
            print(f"world {world} rank {rank}: mll {gp.mll:.6f} (oracle {ref['mll']:.6f}) logdet {gp.logdet:.6f} ({ref['logdet']:.6f}) diag err per block",
                  [f"{e[b*WD:(b+1)*WD].max():.1e}" for b in range(gp.nblk)], "alpha err per block", [f"{ea[b*WD:(b+1)*WD].max():.1e}" for b in range(gp.nblk)], flush=True)
        except BaseException as ex:
            import traceback; traceback.print_exc(); shared.barrier.abort()
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]; [t.join() for t in ts]
PY
echo "== blocked / sharded GPU tests"
timeout 1500 python -m pytest tests/test_gpu_dist.py -q -m gpu --durations=6 > $O/c_tests_dist.log 2>&1; grep -v amdgpu $O/c_tests_dist.log | tail -25
echo "== dense path: one stream set per factorisation"
for cfg in "GPMI_CUMASK=0" "GPMI_X=default"; do
  env $cfg timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary c2 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$cfg', 'n50000 ms %.1f frac %.3f | c2 ms %.2f frac %.3f' % (j['ms_per_step'], j['roofline']['frac'], j['c2']['ms_per_step'], j['c2']['roofline_frac']))"
done 2>&1 | tee $O/c_ab.log
echo "== blocked code path on one rank (world-1 overhead): N=50000 + c2 + c4 size"
timeout 600 python bench.py --mode sharded --steps 5 --warmup 2 --no-cpu-baseline --secondary c2,c4 2> $O/c_sharded.err | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('sharded world 1: n50000 ms %.1f frac %.3f mll %.6f | c2 ms %.2f | c4 %s' % (j['ms_per_step'], j['roofline']['frac'], j['config']['mll'], j['c2']['ms_per_step'], json.dumps(j.get('c4_single_gpu'))))
print(json.dumps(j['stage_ms_per_step']))" 2>&1 | tee $O/c_sharded.log
