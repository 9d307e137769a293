"""Isolated trailing update (lower region, K = 2048) as the look-ahead launches it: plain stream with 16 slots / 8 CUs left free, and
the CU-masked update stream (248 CUs) — both kernels (a GPMI_TOOLS build; GPMI_BENCH_STREAM, GPMI_UPDATE256_ON_MASKED)."""
import os
import subprocess
import sys

if len(sys.argv) > 1:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussianprocesses.jl_amd"))
    import gpmi355x as g

    c = g.Context.default(0)
    for (m, k) in ((30720, 2048), (23040, 1024)):
        for v in (0, 256):
            ms = c.bench_gemm(m, m, k, 1, v, 3)
            print(f"{sys.argv[1]:8s} M=N={m} K={k} {'256x128' if v else '128x128'} {ms:8.3f} ms {(m * (m + 1.0) * k) / ms / 1e9:6.1f} TF", flush=True)
else:
    for mode in ("full", "reserve", "masked"):
        env = dict(os.environ, GPMI_BENCH_STREAM=mode, GPMI_UPDATE256_ON_MASKED="1")
        subprocess.call([sys.executable, __file__, mode], env=env)
