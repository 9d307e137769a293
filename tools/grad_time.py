"""Wall time of update_mll! and update_dmll! at the C2 size (N=20000, d=8, SEArd) — the §8f-1 row."""
import math
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gaussianprocesses.jl_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gpmi355x as g
from bench import synthetic_inputs

for n in (5000, 20000, 50000):
    x, y, xs = synthetic_inputs(n, 8, 16)
    ll = [math.log(0.5) + 0.05 * k for k in range(8)]
    gp = g.GP(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1))
    gp.update_dmll()
    t0 = time.perf_counter(); gp.update_mll(); t1 = time.perf_counter(); gp.update_dmll(); t2 = time.perf_counter()
    print(f"N={n}: update_mll {1e3*(t1-t0):.1f} ms, update_dmll {1e3*(t2-t1):.1f} ms, dmll[:3]={gp.dmll[:3]}")
    prof = gp.ctx.profile() if hasattr(gp.ctx, "profile") else None
