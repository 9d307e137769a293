"""Freeze golden vectors for the 12 kernels of the reference's perf benchmark
(perf/benchmarks/benchmark_julia.jl:10-24: GP(X, Y, MeanConst(0.0), kern, 0.3)) on the
reference's own deterministic input fixture perf/benchmarks/simdata.csv (first N rows).

The Julia reference cannot run in the build container, so the OUTPUTS stored here come
from the CPU oracle (oracle/gp_oracle.py), which is itself pinned against scikit-learn and
the reference's relational tests (tests/test_oracle.py).  They freeze today's numbers so
that later rounds cannot drift silently; they are not reference-produced goldens
("parity unpinned by reference outputs").

Run from the repo root (needs /root/reference):  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gp_oracle as G  # noqa: E402

N, P = 400, 16
SE, M12, RQ = ("se_iso", 0.3, 0.3), ("mat12_iso", 0.3, 0.3), ("rq_iso", 0.3, 0.3, 0.3)
KERNS = {
    "se": SE,
    "mat12": M12,
    "rq": RQ,
    "se+rq": ("sum", SE, RQ),
    "se+mat12": ("sum", SE, M12),
    "se*rq": ("prod", SE, RQ),
    "se*mat12": ("prod", SE, M12),
    "se+mat12+rq": ("sum", ("sum", SE, M12), RQ),
    "(se+mat12)*rq": ("prod", ("sum", SE, M12), RQ),
    "mask(se,[1])": ("masked", SE, [0]),
    "mask(se,[1])+mask(rq,[2:10])": ("sum", ("masked", SE, [0]), ("masked", RQ, list(range(1, 10)))),
    "fix(se,σ)": ("fixed", SE, [0]),
}


def main():
    csv = "/root/reference/perf/benchmarks/simdata.csv"
    data = np.loadtxt(csv, delimiter=",", skiprows=1)
    x = np.ascontiguousarray(data[:N, :10].T)  # d x N
    y = data[:N, 10].copy()
    xpred = np.ascontiguousarray(data[N:N + P, :10].T)
    out = {"x": x, "y": y, "xpred": xpred, "log_noise": np.array(0.3), "mean_const": np.array(0.0),
           "names": np.array(list(KERNS))}
    for i, (name, spec) in enumerate(KERNS.items()):
        fit = G.update_mll(spec, x, y, 0.3, ("const", 0.0))
        mu, s2 = G.predict_f(spec, x, fit, xpred, ("const", 0.0))
        out[f"mll_{i}"] = np.array(fit["mll"])
        out[f"alpha_{i}"] = fit["alpha"]
        out[f"mu_{i}"] = mu
        out[f"s2_{i}"] = s2
        print(f"{name:32s} mll = {fit['mll']:.12f}")
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "simdata_bench_kernels.npz"), **out)


if __name__ == "__main__":
    main()
