#!/bin/bash
# round 5, call H: the adaptive U2a split (blocked.cpp): the overlap test with its new model, the multi-rank GPU tests, the rehearsal line
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
timeout 900 python -m pytest tests/test_gpu_dist.py -q -m gpu -s -k "cu_partitions_and_injected" 2>&1 | grep -E "injected-latency|passed|failed|Error|assert" | cut -c1-1500
timeout 900 python -m pytest tests/test_gpu_dist.py -q -m gpu -x 2>&1 | tail -4
echo "== two partitions, N = 32768 (round 4: 332 - 344 ms; the whole device dense: 233)"
timeout 600 python - <<'PY'
import math, os, sys, time
sys.path.insert(0, "gaussianprocesses.jl_amd")
import numpy as np
import gpmi355x as g
from gpmi355x import dist as gd
n, d = 32768, 8
rng = np.random.default_rng(17)
x = rng.uniform(size=(d, n)); y = np.sin(2 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
ll = [math.log(0.5) + 0.05 * k for k in range(d)]
ctx = g.Context(devices=[256, 512])
gp = gd.ShardedGPE(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1), ctx=ctx, block=1024)
ts = []
for r in range(4):
    t0 = time.perf_counter(); gp.update_mll(); ts.append(time.perf_counter() - t0)
print("two partitions n=32768: fit %.1f ms (min of 4), mll %.4f" % (1e3 * min(ts), gp.mll))
del gp; ctx.close()
gp = g.GP(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1))
ts = []
for r in range(4):
    t0 = time.perf_counter(); gp.update_mll(); ts.append(time.perf_counter() - t0)
print("whole device dense n=32768: fit %.1f ms, mll %.4f" % (1e3 * min(ts), gp.mll))
PY
echo "== rehearsal"
timeout 900 python bench.py --gpus 2 --dry-run-one-gpu --n 16384 --c4-n 24576 --steps 2 --warmup 1 2>/dev/null | tail -1 | cut -c1-400
} > $O/r05_h_adaptive_u2a.log 2>&1
cat $O/r05_h_adaptive_u2a.log
