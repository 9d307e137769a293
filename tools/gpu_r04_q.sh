#!/bin/bash
# round 4, call Q: the next diagonal block from the owner's own rows, ahead of the wait for the panel exchange — blocked tests, then the
# two-partition fit with and without injected exchange latency
mkdir -p gpurun_out; O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_dist.py -q -m gpu -x > $O/r04_q_tests_dist.log 2>&1; tail -1 $O/r04_q_tests_dist.log
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
def fit():
    ts = []
    for rep in range(3):
        t0 = time.perf_counter(); gp.update_mll(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
t0 = fit()
out = [f"no delay {t0:.0f}"]
for what, name in ((2, "exchange"), (1, "broadcast")):
    os.environ["GPMI_TEST_COMM_DELAY_ON"] = str(what); os.environ["GPMI_TEST_COMM_DELAY_US"] = "5000"
    t = fit(); out.append(f"{name}+5ms {t:.0f} (exposed {(t - t0) / (5.0 * (n // 1024 - (1 if what == 2 else 0))):.2f})")
print(f"two partitions n={n}: " + ", ".join(out) + f", mll {gp.mll:.4f}")
PY
for n in 32768 65536; do timeout 300 python /tmp/part_fit.py $n 2>&1 | grep partitions; done | tee $O/r04_q_partitions.log
for n in 50000 20000; do
timeout 300 python bench.py --n $n --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --mode sharded 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('n=$n one rank blocked: ms %.1f fit %.1f predict %.1f frac %.3f' % (j['ms_per_step'], j['fit_only_ms_per_step'], j['predict_only_ms_per_step'], j['roofline']['frac']))"
done | tee -a $O/r04_q_partitions.log
