"""Ablation of the trailing-update kernel in isolation (tools only)."""
import sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussianprocesses.jl_amd"))
import gpmi355x as g

c = g.Context.default(0)
names = {0: "product", 1: "no C read", 2: "no epilogue", 4: "no global loads in loop", 6: "no gl loads + no epilogue",
         8: "no DPP", 14: "no DPP/gl/epi", 22: "no LDS reads/gl/epi", 30: "MFMA only (no DPP/LDS/gl/epi)", 32: "setprio around MFMA"}
variants = [int(a) for a in sys.argv[1:]] or [0, 32, 0, 32]
for (M, N, K, lower) in [(19840, 19840, 256, 1), (20033, 20032, 256, 1), (8192, 8192, 256, 1), (8192, 8192, 2048, 0)]:
    entries = (0.5 * N * (N + 1) + (M - N) * N) if lower else M * N
    fl = 2.0 * entries * K
    print(f"--- M={M} N={N} K={K} lower={lower}  ({fl/1e9:.1f} GFLOP)")
    for v in variants:
        ms = min(c.bench_gemm(M, N, K, lower, v, 3) for _ in range(2))
        print(f"  variant {v:3d} {names[v % 64] + (' [odd-stride ld]' if v >= 64 else ''):44s} {ms:8.3f} ms  {fl/ms/1e9:7.2f} TFLOP/s")
