#!/bin/bash
# round 6, call J: a model sharded over the two CU partitions of the one GPU (N = 32 768: 315 ms in round 5, bar 290) next to the whole device,
# and the multi-GPU bench line rehearsed on one GPU (bench.py's world > 1 branch was rewritten this round: primary line persisted first,
# emit under a lock, exit code on parity failure).
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
echo "== two partitions, N = 32768 / 65536"
timeout 600 python - <<'PY'
import math, os, sys, time
sys.path.insert(0, "gaussianprocesses.jl_amd")
import numpy as np
import gpmi355x as g
from gpmi355x import dist as gd
for n in (32768, 65536):
    d = 8
    rng = np.random.default_rng(17)
    x = rng.uniform(size=(d, n)); y = np.sin(2 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
    ll = [math.log(0.5) + 0.05 * k for k in range(d)]
    ctx = g.Context(devices=[256, 512])
    gp = gd.ShardedGPE(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1), ctx=ctx, block=1024)
    ts = []
    for r in range(4):
        t0 = time.perf_counter(); gp.update_mll(); ts.append(time.perf_counter() - t0)
    print("two partitions n=%d: fit %.1f ms (min of 4), mll %.4f" % (n, 1e3 * min(ts), gp.mll), flush=True)
    del gp; ctx.close()
    gp = g.GP(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1))
    ts = []
    for r in range(4):
        t0 = time.perf_counter(); gp.update_mll(); ts.append(time.perf_counter() - t0)
    print("whole device dense n=%d: fit %.1f ms, mll %.4f" % (n, 1e3 * min(ts), gp.mll), flush=True)
    del gp
PY
echo "== rehearsal of the multi-GPU line on one GPU"
rm -f $O/bench_gpus2*.json
timeout 900 python bench.py --gpus 2 --dry-run-one-gpu --n 16384 --c4-n 24576 --steps 2 --warmup 1 2>$O/r06_j_rehearsal.err | tail -1 > $O/r06_j_multi_gpu_line_rehearsal_one_gpu.json
python - <<'PY'
import json, os
j = json.load(open("gpurun_out/r06_j_multi_gpu_line_rehearsal_one_gpu.json"))
print("  keys:", sorted(j.keys()))
print("  value %.3f fits/s, n_gpus %s, parity %s, c4_sharded parity %s, per_step_ms %s" % (j["value"], j["n_gpus"], j.get("parity", {}).get("ok"), (j.get("c4_sharded") or {}).get("parity", {}).get("ok"), "yes" if j.get("per_step_ms") else "no"))
print("  persisted:", sorted(f for f in os.listdir("gpurun_out") if f.startswith("bench_gpus2")))
PY
tail -3 $O/r06_j_rehearsal.err | grep -v amdgpu
} > $O/r06_j.log 2>&1
cat $O/r06_j.log
