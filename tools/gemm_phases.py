import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussianprocesses.jl_amd"))
import gpmi355x as g
c = g.Context.default(0)
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else (0, 128)
shapes = ((19840, 19840, 256, 1),) if len(sys.argv) > 1 and sys.argv[1] == "one" else ((19840, 19840, 256, 1), (19840, 19840, 1024, 1), (1024, 256, 256, 0), (19840, 256, 256, 0), (4992, 4992, 256, 1))
for (m, n, k, lower) in shapes:
    for v in variants:
        ms = c.bench_gemm(m, n, k, lower, v, 3)
        fl = (n * (n + 1.0) + 2.0 * (m - n) * n) * k if lower else 2.0 * m * n * k
        print(f"M={m} N={n} K={k} lower={lower} variant={v}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TF", flush=True)
