"""Library yardstick for the trailing-update shape: C[N x N] -= A[N x K] A' at K = 256 / 1024 through torch (rocBLAS / hipBLASLt)."""
import time, torch
dev = torch.device("cuda", 0)
for n, k in ((19840, 256), (19840, 1024), (9984, 256)):
    A = torch.randn(n, k, dtype=torch.float64, device=dev)
    C = torch.randn(n, n, dtype=torch.float64, device=dev)
    for _ in range(2): C.addmm_(A, A.t(), beta=1.0, alpha=-1.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 5
    for _ in range(reps): C.addmm_(A, A.t(), beta=1.0, alpha=-1.0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"torch addmm_ fp64 N={n} K={k}: {dt*1e3:.3f} ms  full-square {2.0*n*n*k/dt/1e12:.1f} TFLOP/s  (lower-triangle-equivalent time would be {dt*1e3/2:.3f} ms)")
