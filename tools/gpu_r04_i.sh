#!/bin/bash
# round 4, call I: the sharded fit over two CU partitions after "the chain waits for the diagonal block only"; U2a share sweep
mkdir -p gpurun_out; O=gpurun_out
cat > /tmp/part_fit.py <<'PY'
import math, os, sys, time
import numpy as np
root = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, os.path.join(root, "gaussianprocesses.jl_amd")); sys.path.insert(0, root)
import gpmi355x as g
from gpmi355x import dist as gd
n, d = int(sys.argv[1]), 8
rng = np.random.default_rng(17)
x = rng.uniform(size=(d, n)); y = np.sin(2 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
ll = [math.log(0.5) + 0.05 * k for k in range(d)]
ctx = g.Context(devices=[256, 512])
gp = gd.ShardedGPE(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1), ctx=ctx, block=1024)
ts = []
for rep in range(4):
    t0 = time.perf_counter(); gp.update_mll(); ts.append(time.perf_counter() - t0)
print(f"two partitions n={n} U2A={os.environ.get('GPMI_BLOCKED_U2A', '4')}: fit {min(ts) * 1e3:.1f} ms, mll {gp.mll:.4f}")
PY
for u in 4 3 2; do GPMI_BLOCKED_U2A=$u timeout 300 python /tmp/part_fit.py 32768 2>&1 | grep partitions; done | tee $O/r04_i_partitions.log
GPMI_BLOCKED_U2A=4 timeout 300 python /tmp/part_fit.py 50000 2>&1 | grep partitions | tee -a $O/r04_i_partitions.log
