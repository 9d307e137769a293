import math, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gaussianprocesses.jl_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import gpmi355x as g
from bench import synthetic_inputs
for n in (16384, 16320, 8192):
    x, y, xs = synthetic_inputs(n, 8, 256)
    gp = g.GP(x, y, g.MeanZero(), g.SEArd([math.log(0.5) + 0.05 * k for k in range(8)], 0.0), math.log(0.1))
    t0 = time.perf_counter(); gp.update_mll(); t1 = time.perf_counter(); mu, v = gp.predict_f(xs); t2 = time.perf_counter()
    print(f"N={n}: update_mll {1e3*(t1-t0):.1f} ms ({n**3/3/(t1-t0)/1e12:.1f} TF eq), predict {1e3*(t2-t1):.1f} ms, mll {gp.mll:.6f}")
