#!/usr/bin/env python
"""bench.py — GP fits/sec (update_mll! + predict_f) on MI355X, BASELINE.json's metric.

One "step" = the reference's steady-state unit of work:
    set_params!(gp, hyp)  ->  update_mll!(gp)  ->  predict_f(gp, xpred)      (full_cov=false)
with x already resident in HBM (uploaded once by GP(), as fit! does) and the hyper-parameters
perturbed every step so nothing can be cached.  Workload at N=1: BASELINE.json configs[1]
(N=20000, d=8, SEArd, fp64, MeanZero, P=1024 test points, SURVEY §8d hyper-parameters).

N > 1 (one process per GPU, torchrun): by default every rank runs its own fit (the metric's unit is a fit, the
units are independent: weak scaling, no data-path collective) and, as an extra "sharded_leg", ONE fit of the same
size is also run row-block sharded over all GPUs with the RCCL panel all-gather (gpmi355x.dist); `--mode sharded`
makes that the measured workload instead (strong scaling).  The extra leg runs AFTER the JSON line has been printed and
reports on stderr / gpurun_out/sharded_leg_<N>.json.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     — the dominant kernel (Cholesky trailing update, MFMA-bound): algorithmic flops per
                 launch / mean launch duration, measured live with HIP events on the library's
                 stream over the timed region (gpmi_profile_*).
  cpu_baseline — the CPU oracle (a port: the Julia reference cannot run here) timed on this host
                 on a bounded sample and scaled to the bench size stage by stage.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "gaussianprocesses.jl_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix (v_mfma_f64_16x16x4_f64): 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: Peak FP32 (matrix)


def _pmc_traffic(args, n, d, p):
    """HBM bytes per launch of the roofline kernel.  PMC counters cannot be read from inside the timed process, so the
    number comes from the committed summary of the separate rocprofv3 --pmc passes over THIS command
    (tools/gpu_pmc_bench.sh -> profiles/r01_bench_pmc_hbm.json: FETCH_SIZE and WRITE_SIZE in their own passes, the
    gfx950 x2 read correction of MI355X_MICROARCH.md applied).  It only applies to the default workload; anything
    else reports null."""
    path = os.path.join(ROOT, "profiles", "r01_bench_pmc_hbm.json")
    default = (n, d, p, args.dtype) == (20000, 8, 1024, "f64")
    if not default or not os.path.exists(path):
        return {"traffic": None}
    try:
        j = json.load(open(path))
        return {"traffic": j["traffic_bytes_per_launch"], "traffic_unit": "HBM bytes per launch (PMC, profiles/r01_bench_pmc_hbm.json)"}
    except Exception:
        return {"traffic": None}


def synthetic_inputs(n, d, p, seed=20240501):
    """SURVEY.md §8(d) inputs (same generator as oracle.gp_oracle.synthetic_inputs)."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(0.0, 1.0, size=(d, n))
    f = np.sin(2.0 * np.pi * x).sum(axis=0) / d
    y = f + 0.1 * rng.standard_normal(n)
    xpred = rng.uniform(0.0, 1.0, size=(d, p))
    return x, y, xpred


def cpu_baseline(n_bench, d, p, ll, n_sample):
    """Oracle timed on the host: cov! as the reference's single-threaded scalar loop (C), LAPACK
    dpotrf/dpotrs/dtrsm on all cores; scaled from n_sample to n_bench per stage (N^2 / N^3)."""
    import scipy.linalg as sla

    from oracle import c_oracle
    from oracle import gp_oracle as G

    x, y, xs = G.synthetic_inputs(n_sample, d, p)
    spec = ("se_ard", ll, 0.0)
    t0 = time.perf_counter()
    K = c_oracle.assemble(spec, x, math.log(0.1))
    t1 = time.perf_counter()
    U, info = sla.lapack.dpotrf(K, lower=0, clean=0, overwrite_a=1)
    assert info == 0
    t2 = time.perf_counter()
    alpha = sla.cho_solve((U, False), y)
    mll = -(float(y @ alpha) + 2.0 * np.sum(np.log(np.diag(U))) + G.LOG2PI * n_sample) / 2.0
    t3 = time.perf_counter()
    Kc = c_oracle.cov(spec, x, xs)
    mu = Kc.T @ alpha
    Lck = sla.solve_triangular(U, Kc, trans="T", lower=False, overwrite_b=True)
    s2 = np.maximum(1.0 - np.sum(Lck * Lck, axis=0), 0.0)
    t4 = time.perf_counter()
    assert np.isfinite(mll) and np.all(np.isfinite(mu)) and np.all(np.isfinite(s2))
    r = n_bench / n_sample
    t_cov, t_chol, t_solve, t_pred = t1 - t0, t2 - t1, t3 - t2, t4 - t3
    est = t_cov * r**2 + t_chol * r**3 + t_solve * r**2 + t_pred * r**2
    return {
        "value": 1.0 / est,
        "unit": "GP fits/sec",
        "cores": os.cpu_count(),
        "kind": "port",
        "sample": (f"oracle fit+predict measured at N={n_sample} d={d} P={p} "
                   f"(cov! 1-thread C loop {t_cov:.2f}s, dpotrf {t_chol:.2f}s, dpotrs+mll {t_solve:.2f}s, "
                   f"predict {t_pred:.2f}s = {1.0/(t4-t0):.4f} fits/s), scaled to N={n_bench} per stage (N^2/N^3)"),
    }


def _main_json(args, world, elapsed, gp, n, d, p, fl_syrk, ms_syrk, n_syrk, ms_cov, by_cov, ms_pan, ms_sol, ms_pre, t_build,
               scaling, sharded=False):
    peak = FP64_MFMA_PEAK_TFLOPS if args.dtype == "f64" else FP32_MFMA_PEAK_TFLOPS
    achieved = (fl_syrk / max(ms_syrk, 1e-9)) * 1e-9  # flop/ms -> TFLOP/s
    nfits = args.steps if sharded else world * args.steps
    if world == 1:
        par = "single GPU"
    elif sharded:
        par = f"ONE fit row-block sharded over {world} GPUs (block-cyclic 256-row blocks, RCCL panel all-gather per step)"
    else:
        par = f"{world} independent fits, one per GPU (the metric's unit is a fit: no data-path collective)"
    return {
        "metric": "GP fits/sec (update_mll!+predict_f)",
        "value": nfits / elapsed,
        "unit": "GP fits/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {
            "workload": f"N={n}, d={d}, SEArd + MeanZero, {args.dtype}, P={p} test points, full_cov=false "
                        "(BASELINE.json configs[1])",
            "parallelism": par,
            "mll": gp.mll,
        },
        "roofline": {
            "kernel": "gemm_nt_kernel<T, 0, 4> (Cholesky trailing update, 128x128 tiles, K=256, v_mfma_f64_16x16x4)",
            "bound": "mfma",
            "achieved": achieved,
            "peak": peak,
            "unit": "TFLOP/s",
            "frac": achieved / peak,
            **_pmc_traffic(args, n, d, p),
            "launches": n_syrk,
            "avg_launch_ms": ms_syrk / max(n_syrk, 1),
            "algorithmic_flops_per_launch": fl_syrk / max(n_syrk, 1),
            # C tile read + write (8 B each) per 2 * 256 flops of an entry; the 256-column panel itself is read once
            "algorithmic_bytes_per_launch": fl_syrk / max(n_syrk, 1) / (2.0 * 256.0) * 2 * (8 if args.dtype == "f64" else 4),
        },
        "stage_ms_per_step": {
            "cov": ms_cov / args.steps,
            "cov_GBps": (by_cov / max(ms_cov, 1e-9)) * 1e-6,
            "chol_trailing_update": ms_syrk / args.steps,
            "panel_potf2_trsm_update": ms_pan / args.steps,
            "alpha_solve_mll": ms_sol / args.steps,
            "predict": ms_pre / args.steps,
            "note": "per-class sums of HIP-event intervals; the panel chain runs on a side stream UNDER the trailing update "
                    "(look-ahead), so the classes overlap and do not add up to ms_per_step",
        },
        "first_fit_incl_upload_s": t_build,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=20000)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--p", type=int, default=1024)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--cpu-sample-n", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="replicas", choices=["replicas", "sharded"],
                    help="N>1: independent fits per GPU (default, the metric's unit is a fit) or ONE fit row-block "
                         "sharded over the GPUs with the RCCL panel all-gather (gpmi355x.dist)")
    ap.add_argument("--no-sharded-leg", action="store_true", help="N>1 replicas mode: skip the extra sharded measurement")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: no GPU visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import gpmi355x as g

    n, d, p = args.n, args.d, args.p
    np_dt = np.float64 if args.dtype == "f64" else np.float32
    x, y, xpred = synthetic_inputs(n, d, p)
    ll = [math.log(0.5) + 0.05 * k for k in range(d)]
    log_noise = math.log(0.1)
    ctx = g.Context.default(local_rank)
    t_build0 = time.perf_counter()
    sharded = args.mode == "sharded"
    if sharded:
        from gpmi355x import dist as gd

        comm = gd.TorchDistComm() if dist is not None else gd.SingleComm()
        gp = gd.ShardedGPE(x, y, g.MeanZero(), g.SEArd(ll, 0.0), log_noise, dtype=np_dt, comm=comm, ctx=ctx)
    else:
        gp = g.GP(x, y, g.MeanZero(), g.SEArd(ll, 0.0), log_noise, dtype=np_dt, ctx=ctx)  # uploads x, first fit
    t_build = time.perf_counter() - t_build0
    base = np.asarray(gp.get_params())

    def step(i):
        gp.set_params(base + 1e-3 * ((i % 7) + 1) * np.where(np.arange(len(base)) == 0, 0.0, 1.0))
        gp.update_mll()
        return gp.predict_f(xpred)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    ctx.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        mu, s2 = step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    n_syrk, ms_syrk, fl_syrk = ctx.profile_get(g._lib.PROF_SYRK)
    n_cov, ms_cov, by_cov = ctx.profile_get(g._lib.PROF_COV)
    n_pan, ms_pan, fl_pan = ctx.profile_get(g._lib.PROF_PANEL)
    n_sol, ms_sol, _ = ctx.profile_get(g._lib.PROF_SOLVE)
    n_pre, ms_pre, _ = ctx.profile_get(g._lib.PROF_PREDICT)
    ctx.profile_enable(False)

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    assert np.all(np.isfinite(mu)) and np.all(np.isfinite(s2)) and math.isfinite(gp.mll)

    out = None
    if rank == 0:
        out = _main_json(args, world, elapsed, gp, n, d, p, fl_syrk, ms_syrk, n_syrk, ms_cov, by_cov, ms_pan, ms_sol, ms_pre,
                         t_build, "strong" if sharded else "weak", sharded)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, d, p, ll, min(args.cpu_sample_n, n))

    # The ONE JSON line of the contract goes out first: nothing after this point can cost the measurement.
    if rank == 0:
        print(json.dumps(out), flush=True)

    # Extra leg (N > 1, replicas mode): the SAME workload as ONE fit row-block sharded over all GPUs, so that the RCCL
    # panel all-gather path is exercised and timed on real multi-GPU hardware (this build's containers have one GPU: the
    # path is covered by gloo / virtual-rank tests only).  Reported on stderr and in gpurun_out/sharded_leg_<N>.json,
    # never on stdout; a hang is cut by the watchdog.
    if world > 1 and not sharded and not args.no_sharded_leg:
        import threading

        done = threading.Event()

        def report(leg):
            if rank == 0:
                sys.stderr.write("[sharded_leg] " + json.dumps(leg) + "\n")
                sys.stderr.flush()
                try:
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    with open(os.path.join(ROOT, "gpurun_out", f"sharded_leg_{world}.json"), "w") as fh:
                        json.dump(leg, fh)
                except OSError:
                    pass

        def watchdog():
            if not done.wait(240.0):
                report({"error": "no result within 240 s (watchdog)"})
                os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            from gpmi355x import dist as gd

            sgp = gd.ShardedGPE(x, y, g.MeanZero(), g.SEArd(ll, 0.0), log_noise, dtype=np_dt, comm=gd.TorchDistComm(), ctx=ctx)
            barrier()
            ts = time.perf_counter()
            nrep = 2
            for i in range(nrep):
                sgp.set_params(base + 1e-3 * (i + 1) * np.where(np.arange(len(base)) == 0, 0.0, 1.0))
                sgp.update_mll()
                smu, ss2 = sgp.predict_f(xpred)
            barrier()
            dt = (time.perf_counter() - ts) / nrep
            leg = {"workload": f"ONE fit+predict of N={n} row-block sharded over {world} GPUs (RCCL panel all-gather)",
                   "fits_per_sec": 1.0 / dt, "ms_per_step": 1e3 * dt, "mll": sgp.mll,
                   "finite": bool(np.all(np.isfinite(smu)) and np.all(np.isfinite(ss2)))}
        except Exception as e:  # noqa: BLE001
            leg = {"error": repr(e)[:300]}
        done.set()
        report(leg)

    if dist is not None:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
    if world > 1:
        sys.stdout.flush()
        os._exit(0)  # skip interpreter teardown of RCCL / HIP objects in multi-process runs


if __name__ == "__main__":
    main()
