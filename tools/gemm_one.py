import sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussianprocesses.jl_amd"))
import gpmi355x as g
c = g.Context.default(0)
v = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ms = c.bench_gemm(19840, 19840, 256, 1, v, 3)
print("variant", v, ms, "ms", 100.8 / ms, "TF")
