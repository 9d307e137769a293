#!/bin/bash
# round 3, call N: per-step times at N = 50000 / 20000 under the three instrumentation modes (none, roofline kernel only, every class)
mkdir -p gpurun_out; O=gpurun_out
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu | tee $O/n_steps.log
import math, sys, time, os
sys.path.insert(0, "gaussianprocesses.jl_amd"); sys.path.insert(0, ".")
import numpy as np
import gpmi355x as g
rng = np.random.default_rng(20240501)
ctx = g.Context.default(0)
for n in (50000, 20000):
    d = 8
    x = rng.uniform(size=(d, n)); y = np.sin(2 * np.pi * x).sum(axis=0) / d + 0.1 * rng.standard_normal(n); xs = rng.uniform(size=(d, 1024))
    ll = [math.log(0.5) + 0.05 * k for k in range(d)]
    gp = g.GP(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1), ctx=ctx)
    base = np.asarray(gp.get_params())
    def step(i):
        gp.set_params(base + 1e-3 * ((i % 7) + 1) * np.where(np.arange(len(base)) == 0, 0.0, 1.0))
        t0 = time.perf_counter(); gp.update_mll(); t1 = time.perf_counter(); gp.predict_f(xs); t2 = time.perf_counter()
        return 1e3 * (t1 - t0), 1e3 * (t2 - t1)
    for mode, kw in (("none", None), ("no-chain", dict(skip_chain=True)), ("all", dict()), ("syrk-only", dict(only=g._lib.PROF_SYRK)), ("no-chain", dict(skip_chain=True))):
        if kw is None: ctx.profile_enable(False)
        else: ctx.profile_enable(True, **kw)
        ts = [step(i) for i in range(7)]
        ctx.profile_enable(False)
        print(f"N={n} events {mode:9s}: fit " + " ".join(f"{a:6.1f}" for a, b in ts) + " | predict " + " ".join(f"{b:5.1f}" for a, b in ts), flush=True)
    del gp
PY
