"""Library yardsticks on the same GPU (NOT part of the product path): rocBLAS/hipBLASLt dgemm and
hipSOLVER/MAGMA potrf through torch, to know what the vendor stack reaches in FP64 on this chip."""
import time
import torch

dev = torch.device("cuda")
for n in (4096, 8192):
    a = torch.randn(n, n, dtype=torch.float64, device=dev)
    b = torch.randn(n, n, dtype=torch.float64, device=dev)
    torch.mm(a, b); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        c = torch.mm(a, b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"torch.mm fp64 n={n}: {dt*1e3:.2f} ms  {2*n**3/dt/1e12:.2f} TFLOP/s")
    del a, b, c
for n in (8192, 20000):
    a = torch.randn(n, 64, dtype=torch.float64, device=dev)
    k = a @ a.T + n * torch.eye(n, dtype=torch.float64, device=dev)
    torch.linalg.cholesky(k[:512, :512]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    l = torch.linalg.cholesky(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"torch.linalg.cholesky fp64 n={n}: {dt*1e3:.1f} ms  {n**3/3/dt/1e12:.2f} TFLOP/s")
    del a, k, l
