import sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussianprocesses.jl_amd"))
import gpmi355x as g
c = g.Context.default(0)
v = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for (m, k) in ((19840, 256), (9984, 256), (4992, 256)):
    ms = c.bench_gemm(m, m, k, 1, v, 5)
    print(os.environ.get("GPMI_PANEL_CUS"), os.environ.get("GPMI_MAIN_MASKED"), "M=N=", m, "K=", k, f"{ms:.3f} ms", f"{(m*(m+1.0)*k)/ms/1e9:.1f} TF")
