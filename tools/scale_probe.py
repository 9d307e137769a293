"""Timing / sanity of the other BASELINE.json configs on one GPU (tools only):
C3  N=50000 d=8 (SEArd+Mat52Iso)+Noise fp64;  fp32 at N=100000 and (if memory allows) N=200000 d=16 SEArd (C4's size)."""
import math, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussianprocesses.jl_amd"))
import numpy as np
import gpmi355x as g

def synth(n, d, p, seed=20240501):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0.0, 1.0, size=(d, n)); y = np.sin(2*np.pi*x).sum(axis=0)/d + 0.1*rng.standard_normal(n)
    return x, y, rng.uniform(0.0, 1.0, size=(d, p))

def run(name, n, d, kern, dtype, p=1024, reps=2):
    x, y, xs = synth(n, d, p)
    t0 = time.perf_counter(); gp = g.GP(x, y, g.MeanZero(), kern, math.log(0.1), dtype=dtype); t1 = time.perf_counter()
    ts = []
    for i in range(reps):
        hyp = np.asarray(gp.get_params()); hyp[1:] += 1e-3
        a = time.perf_counter(); gp.set_params(hyp); gp.update_mll(); b = time.perf_counter(); mu, s2 = gp.predict_f(xs); c = time.perf_counter()
        ts.append((b - a, c - b))
    fit = min(t[0] for t in ts); pred = min(t[1] for t in ts)
    fl = n**3 / 3
    print(f"{name}: N={n} d={d} {np.dtype(dtype).name}: ctor {t1-t0:.2f}s  fit {fit*1e3:.1f} ms ({fl/fit/1e12:.1f} TFLOP/s chol-equivalent)  predict(P={p}) {pred*1e3:.1f} ms  "
          f"-> {1/(fit+pred):.3f} fits/s   mll={gp.mll:.6g} finite={bool(np.all(np.isfinite(mu)) and np.all(np.isfinite(s2)) and np.all(s2>=0))}", flush=True)
    del gp

ll8 = [math.log(0.5) + 0.05*k for k in range(8)]
ll16 = [math.log(0.5) + 0.05*k for k in range(16)]
which = sys.argv[1:] or ["c3", "f32_100k"]
if "c2" in which: run("C2", 20000, 8, g.SEArd(ll8, 0.0), np.float64)
if "c3" in which: run("C3", 50000, 8, g.SEArd(ll8, 0.0) + g.Mat52Iso(math.log(0.7), math.log(0.5)) + g.Noise(math.log(0.05)), np.float64)
if "f32_20k" in which: run("fp32 C2-size", 20000, 8, g.SEArd(ll8, 0.0), np.float32)
if "f32_100k" in which: run("fp32", 100000, 16, g.SEArd(ll16, 0.0), np.float32, reps=1)
if "f32_200k" in which: run("C4-size on ONE GPU", 200000, 16, g.SEArd(ll16, 0.0), np.float32, reps=1)
