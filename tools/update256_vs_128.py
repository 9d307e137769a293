"""Isolated timing of the trailing update (lower region of an M x M matrix, K columns): the 256 x 128 kernel (update256.hip,
variant 256 of gpmi_bench_gemm) against the 128 x 128 one (variant 0).  TFLOP/s on the lower region's flops M (M + 1) K."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussianprocesses.jl_amd"))
import gpmi355x as g

c = g.Context.default(0)
dt = sys.argv[1] if len(sys.argv) > 1 else "f64"
for (m, k) in ((46080, 2048), (30720, 2048), (23040, 1024), (15360, 1024), (10240, 512), (8192, 256)):
    row = []
    for v in (0, 256):
        ms = c.bench_gemm(m, m, k, 1, v, 3, dtype=64 if dt == "f64" else 32)
        row.append((ms, (m * (m + 1.0) * k) / ms / 1e9))
    print(f"{dt} M=N={m} K={k}: 128x128 {row[0][0]:8.3f} ms {row[0][1]:6.1f} TF   256x128 {row[1][0]:8.3f} ms {row[1][1]:6.1f} TF   ratio {row[0][0] / row[1][0]:.3f}", flush=True)
