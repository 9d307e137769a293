import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussianprocesses.jl_amd"))
import gpmi355x as g
c = g.Context.default(0)
for (m, n, k, lower) in ((19968, 19968, 256, 1), (19968, 19968, 1024, 1), (4992, 4992, 256, 1)):
    for v in (0, 128, 2):
        ms = c.bench_gemm(m, n, k, lower, v, 3, dtype=32)
        fl = (n * (n + 1.0) + 2.0 * (m - n) * n) * k if lower else 2.0 * m * n * k
        print(f"fp32 M={m} N={n} K={k} lower={lower} variant={v}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TF", flush=True)
