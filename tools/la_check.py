"""Look-ahead bring-up: mll / alpha digest for a few sizes under the current GPMI_* environment."""
import math, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gaussianprocesses.jl_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import gpmi355x as g
from bench import synthetic_inputs
for n in (1000, 3000, 20000):
    x, y, xs = synthetic_inputs(n, 8, 16)
    ll = [math.log(0.5) + 0.05 * k for k in range(8)]
    gp = g.GP(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1))
    t0 = time.perf_counter(); gp.update_mll(); t1 = time.perf_counter()
    print(os.environ.get("GPMI_PANEL_CUS"), os.environ.get("GPMI_LA_MODE"), n, repr(gp.mll), float(np.abs(gp.alpha).sum()), f"{1e3*(t1-t0):.1f} ms")
