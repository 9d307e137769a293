import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussianprocesses.jl_amd"))
import gpmi355x as g
c = g.Context.default(0)
for (m, n, k, lower) in ((1000000, 3840, 256, 0), (100000, 3840, 256, 0), (19840, 3840, 256, 0)):
    for v in (0, 128):
        ms = c.bench_gemm(m, n, k, lower, v, 2)
        print(f"M={m} N={n} K={k} lower={lower} variant={v}: {ms:.2f} ms  {2.0*m*n*k/ms/1e9:.1f} TF", flush=True)
