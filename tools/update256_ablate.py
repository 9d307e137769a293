"""Ablation of update256_kernel in isolation (a GPMI_TOOLS build): variant 256 + bits: 1 no epilogue, 2 no operand DMA after the
prologue, 4 no fragment reads, 8 no slab barrier.  Lower region of M x M, K columns."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussianprocesses.jl_amd"))
import gpmi355x as g

c = g.Context.default(0)
names = {256: "full", 257: "no epilogue", 259: "no epilogue, no DMA", 261: "no epilogue, no fragment reads", 263: "no epilogue, no DMA, no fragment reads",
         271: "bare MFMA loop (also no slab barrier)", 0: "128 x 128 kernel"}
for (m, k) in ((46080, 2048), (23040, 1024)):
    for v in (0, 256, 257, 259, 261, 263, 271):
        ms = c.bench_gemm(m, m, k, 1, v, 3)
        print(f"M=N={m} K={k} {names[v]:45s} {ms:8.3f} ms {(m * (m + 1.0) * k) / ms / 1e9:6.1f} TF", flush=True)
