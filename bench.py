#!/usr/bin/env python
"""bench.py — GP fits/sec (update_mll! + predict_f) on MI355X, BASELINE.json's metric.

One "step" = the reference's steady-state unit of work:
    set_params!(gp, hyp)  ->  update_mll!(gp)  ->  predict_f(gp, xpred)      (full_cov=false)
with x already resident in HBM (uploaded once by GP(), as fit! does) and the hyper-parameters
perturbed every step so nothing can be cached.

Workload (the configuration BASELINE.json's north-star target is quoted on):
    N = 50 000, d = 8, SEArd + MeanZero, fp64, P = 1024 test points, SURVEY §8d hyper-parameters.
The same JSON line carries secondary objects: "c2" (BASELINE configs[1], N = 20 000, a few steps) and, on one GPU,
"c4_single_gpu" (N = 200 000, d = 16, fp32 — the size north_star's multi-GPU target is quoted on — as ONE fit on one
GPU, so that a strong-scaling curve has its single-GPU point).

`--gpus N` (N > 1): one process per GPU.  When no launcher set WORLD_SIZE this script re-executes itself under
`python -m torch.distributed.run --nproc-per-node N` (and fails loudly if the box has fewer GPUs).  The measured
workload is then ONE fit of the same N = 50 000 problem row-block SHARDED over the N GPUs (RCCL panel exchange,
gpmi355x.dist): strong scaling, `value` = fits/s of the whole job.  `--mode replicas` runs N independent fits instead.
The ONE JSON line of a multi-GPU run also carries `parity` (solve residual of the sharded fit on oracle-rebuilt rows), `per_step_ms`
(HIP-event time per block step of chain / broadcast / solve / gather / U1 / U2a / U2b on rank 0 — what a bad scaling curve is diagnosed
from) and "c4_sharded": one N = 200 000, d = 16, fp32 fit sharded over the same GPUs — the configuration north_star's 60 % strong-scaling
target is quoted on — with its own `parity`.  They are computed BEFORE the line is printed, under a watchdog that prints the line
without them if they do not finish in time.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     — the dominant kernel (Cholesky trailing update, MFMA-bound): algorithmic flops per
                 launch / mean launch duration, measured live with HIP events on the library's
                 stream over the timed region (gpmi_profile_*).
  cpu_baseline — the CPU oracle (a port: the Julia reference cannot run here) MEASURED on this host at the bench size
                 (one fit + predict; cov! single-threaded C like the reference's loop — and, beside it, a vectorised-NumPy
                 cov! figure —, LAPACK on the host's cores), with the CPU model, core count and BLAS threads.
  parity       — the device fit + predict_f at the base hyper-parameters against that oracle run (exit code 3 above 1e-5).
Secondary objects carry their own `parity`: c4_single_gpu (solve residual on 512 oracle-rebuilt rows), grad and c5 (analytic
directional derivative against a central difference of the device mll).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "gaussianprocesses.jl_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix (v_mfma_f64_16x16x4_f64): 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: Peak FP32 (matrix)
PMC_RECORD = next((os.path.join("profiles", f"r{r:02d}_bench_pmc_hbm.json") for r in (6, 5, 4)
                   if os.path.exists(os.path.join(ROOT, "profiles", f"r{r:02d}_bench_pmc_hbm.json"))), os.path.join("profiles", "r05_bench_pmc_hbm.json"))


def _pmc_traffic(args, n, d, p):
    """HBM bytes per launch of the roofline kernel.  PMC counters cannot be read from inside the timed process, so the
    number comes from the committed summary of the separate rocprofv3 --pmc passes over THIS command
    (tools/gpu_pmc_bench.sh -> profiles/r04_bench_pmc_hbm.json: FETCH_SIZE and WRITE_SIZE in their own passes, the
    gfx950 x2 read correction of MI355X_MICROARCH.md applied).  It only applies to the default workload; anything
    else reports null."""
    path = os.path.join(ROOT, PMC_RECORD)
    default = (n, d, p, args.dtype) == (50000, 8, 1024, "f64")
    if not default or not os.path.exists(path):
        return {"traffic": None}
    try:
        j = json.load(open(path))
        return {"traffic": j["traffic_bytes_per_launch"], "traffic_unit": f"HBM bytes per launch (PMC, {PMC_RECORD})",
                "traffic_source": f"committed profile {PMC_RECORD} (separate rocprofv3 --pmc passes over this command at the commit named in that file; "
                                  "NOT a measurement of this run - PMC counters cannot be read from inside the timed process)"}
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


# ----------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle, measured
# ----------------------------------------------------------------------------------------------------------------------
def _host_description():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    blas = []
    try:
        import scipy.linalg  # noqa: F401  (loads the BLAS the oracle's LAPACK calls run on)
        from threadpoolctl import threadpool_info

        for lib in threadpool_info():
            blas.append({k: lib.get(k) for k in ("internal_api", "version", "num_threads", "threading_layer") if lib.get(k) is not None})
    except Exception:  # noqa: BLE001
        pass
    nthreads = max([b.get("num_threads", 1) for b in blas if b.get("internal_api") in ("openblas", "mkl", "blis")] or [1])
    return model, blas, nthreads


def _cpu_fit_predict(n, d, p, ll):
    """One oracle fit + predict at size n, timed per stage.  Returns (seconds per stage, total)."""
    import scipy.linalg as sla

    from oracle import c_oracle
    from oracle import gp_oracle as G

    x, y, xs = G.synthetic_inputs(n, d, p)
    spec = ("se_ard", ll, 0.0)
    t0 = time.perf_counter()
    K = c_oracle.assemble(spec, x, math.log(0.1))           # cov! + nugget: the reference's single-threaded loop
    t1 = time.perf_counter()
    U, info = sla.lapack.dpotrf(K, lower=0, clean=0, overwrite_a=1)   # make_posdef! (LAPACK, all cores)
    del K
    assert info == 0
    t2 = time.perf_counter()
    alpha = sla.cho_solve((U, False), y)
    mll = -(float(y @ alpha) + 2.0 * np.sum(np.log(np.diag(U))) + G.LOG2PI * n) / 2.0
    t3 = time.perf_counter()
    Kc = c_oracle.cov(spec, x, xs)
    mu = Kc.T @ alpha
    Lck = sla.solve_triangular(U, Kc, trans="T", lower=False, overwrite_b=True)
    s2 = np.maximum(1.0 - np.sum(Lck * Lck, axis=0), 0.0)
    t4 = time.perf_counter()
    assert np.isfinite(mll) and np.all(np.isfinite(mu)) and np.all(np.isfinite(s2))
    return {"cov": t1 - t0, "dpotrf": t2 - t1, "dpotrs_mll": t3 - t2, "predict": t4 - t3}, t4 - t0, mll, mu, s2


def _numpy_cov_seconds(n, d, ll, rows=2048):
    """SURVEY 8(d)'s SECOND cov! figure: the vectorised-NumPy variant beside the reference-order scalar loop.  K = s2 exp(-r/2) with
    r = |x_i / l - x_j / l|^2 by the BLAS expansion, in row panels of `rows` (the whole N x N would be a second 20 GB matrix); timed on
    min(n, 16384) rows and scaled to N^2 — it is O(N^2) streaming work, stated as scaled."""
    import numpy as _np

    rng = _np.random.default_rng(1)
    x = rng.uniform(size=(n, d)) / _np.exp(_np.asarray(ll))[None, :]
    sq = (x * x).sum(axis=1)
    nrows = min(n, 16384)
    out = _np.empty((rows, n))
    t0 = time.perf_counter()
    for r0 in range(0, nrows, rows):
        r1 = min(r0 + rows, nrows)
        o = out[: r1 - r0]
        _np.matmul(x[r0:r1], x.T, out=o)
        o *= -2.0
        o += sq[r0:r1, None]
        o += sq[None, :]
        _np.maximum(o, 0.0, out=o)
        o *= -0.5
        _np.exp(o, out=o)
    el = time.perf_counter() - t0
    return el * (n / nrows), nrows


def cpu_baseline(n_bench, d, p, ll, budget_s):
    """The oracle MEASURED at the bench size when a probe says it fits `budget_s`, otherwise at N = 20 000 (stated).
    Nothing is scaled: `value` is 1 / (measured seconds) of the run named in `sample`."""
    model, blas, nthreads = _host_description()
    probe_n = min(6000, n_bench)
    st, tot = _cpu_fit_predict(probe_n, d, p, ll)[:2]
    r = n_bench / probe_n
    est = st["cov"] * r**2 + st["dpotrf"] * r**3 + st["dpotrs_mll"] * r**2 + st["predict"] * r**2
    free_gb = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                free_gb = int(line.split()[1]) / 1e6
    except OSError:
        pass
    need_gb = 8.0 * n_bench * n_bench / 1e9 * 1.15
    n_meas = n_bench
    why = ""
    if est > budget_s or (free_gb is not None and free_gb < need_gb):
        n_meas = min(20000, n_bench)
        why = (f" (the bench size N={n_bench} was estimated at {est:.0f} s from a N={probe_n} probe / needs {need_gb:.0f} GB of "
               f"host RAM, over the {budget_s:.0f} s budget: measured at N={n_meas} instead — NOT the bench size)")
    st, tot, mll, mu, s2 = _cpu_fit_predict(n_meas, d, p, ll)
    try:
        np_cov_s, np_rows = _numpy_cov_seconds(n_meas, d, ll)
    except Exception:  # noqa: BLE001
        np_cov_s, np_rows = None, 0
    return {
        "_mu": mu,       # popped by main() before printing: the oracle's predictions for the `parity` object
        "_s2": s2,
        "value": 1.0 / tot,
        "unit": "GP fits/sec",
        "cores": os.cpu_count(),
        "blas_threads": nthreads,
        "kind": "port",
        "n_measured": n_meas,
        "cpu_model": model,
        "blas": blas,
        "stage_s": st,
        # "also report a vectorised-NumPy variant" of cov! — the 12 s single-threaded loop is the REFERENCE's behaviour
        # (kernels.jl:39-50), not the best a CPU can do: with it the fit+predict would take `total_s_with_numpy_cov`
        "cov_numpy_vectorised_s": np_cov_s,
        "cov_numpy_note": (f"K = exp(-|xi - xj|^2 / 2) by the BLAS expansion in 2048-row panels on the host's BLAS threads, timed on {np_rows} rows "
                           f"and scaled to N^2 (streaming O(N^2) work)") if np_cov_s is not None else None,
        "total_s_with_numpy_cov": (tot - st["cov"] + np_cov_s) if np_cov_s is not None else None,
        "fits_per_sec_with_numpy_cov": (1.0 / (tot - st["cov"] + np_cov_s)) if np_cov_s is not None else None,
        "mll": mll,
        "sample": (f"ONE oracle fit+predict MEASURED at N={n_meas} d={d} P={p} SEArd fp64 in {tot:.2f} s: cov! single-threaded C "
                   f"loop (like the reference's) {st['cov']:.2f} s, LAPACK dpotrf {st['dpotrf']:.2f} s on {nthreads} BLAS "
                   f"threads ({os.cpu_count()} logical cores, {model}), dpotrs+mll {st['dpotrs_mll']:.2f} s, predict "
                   f"{st['predict']:.2f} s; nothing scaled" + why),
    }


# ----------------------------------------------------------------------------------------------------------------------
# one timed workload on this rank's GPU
# ----------------------------------------------------------------------------------------------------------------------
def _ll(d):
    return [math.log(0.5) + 0.05 * k for k in range(d)]


def run_workload(g, ctx, n, d, p, dtype, steps, warmup, barrier, comm=None, sharded=False):
    """set_params -> update_mll -> predict_f, `steps` timed.  sharded: ONE fit row-block sharded over comm's ranks (comm None: the
    blocked code path on one rank)."""
    import gc

    gc.collect()
    if "torch" in sys.modules:
        sys.modules["torch"].cuda.empty_cache()
    np_dt = np.float64 if dtype == "f64" else np.float32
    x, y, xpred = synthetic_inputs(n, d, p)
    ll = _ll(d)
    log_noise = math.log(0.1)
    t_build0 = time.perf_counter()
    if sharded:
        from gpmi355x import dist as gd

        gp = gd.ShardedGPE(x, y, g.MeanZero(), g.SEArd(ll, 0.0), log_noise, dtype=np_dt, comm=comm, ctx=ctx)
    else:
        gp = g.GP(x, y, g.MeanZero(), g.SEArd(ll, 0.0), log_noise, dtype=np_dt, ctx=ctx)  # uploads x, first fit
    t_build = time.perf_counter() - t_build0
    base = np.asarray(gp.get_params())
    # the constructor's fit is at the BASE hyper-parameters — the ones the cpu_baseline oracle run uses: keep its mll and
    # predictions so that the JSON line carries full-size parity (main(): `parity`)
    base_mll = gp.mll
    base_alpha = np.asarray(gp.alpha, dtype=np.float64).copy()
    base_mu, base_s2 = gp.predict_f(xpred)

    wall = {"fit": 0.0, "predict": 0.0}

    def step(i):
        gp.set_params(base + 1e-3 * ((i % 7) + 1) * np.where(np.arange(len(base)) == 0, 0.0, 1.0))
        t_a = time.perf_counter()
        gp.update_mll()
        t_b = time.perf_counter()
        out = gp.predict_f(xpred)
        wall["fit"] += t_b - t_a
        wall["predict"] += time.perf_counter() - t_b
        return out

    # Instrumentation inside the timed region (profiles/r03_n_instrumentation.log).  The roofline kernel's launches always carry HIP
    # events (attached to the dispatch: hipExtLaunchKernelGGL start / stop events).  Marker events around EVERY other profiled launch —
    # thousands per fit for the chain kernels — cost 4 ms of a 69 ms step at N = 20 000, so the stage times are taken over the warm-up
    # steps and the timed steps bracket the roofline kernel only.  One measured exception: in the free-slot look-ahead mode (factorisations
    # of >= 12 288 rows since round 6, 18 432 in round 5; N = 20 000 bracketed on the update alone: 81.6 ms per step against 61 – 62 un-instrumented) events on
    # the update alone cost MORE than events everywhere (N = 50 000 fit: none 652, everywhere 657, update
    # only 665 ms — the markers on the side stream's chain kernels evidently help the chain along once the update's dispatch carries a
    # completion signal), so those workloads keep every class bracketed, as in rounds 1 and 2.
    syrk_only = (not sharded) and n < 12288
    how = os.environ.get("GPMI_BENCH_PROFILE", "")   # measurement study: "none" = no event in the timed region (no roofline then), "all", "syrk"
    if how in ("all", "syrk"):
        syrk_only = how == "syrk"
    stage = None
    if warmup > 0:
        ctx.profile_enable(True)
    for i in range(warmup):
        step(i)
    if warmup > 0:
        stage = {name: ctx.profile_get(getattr(g._lib, "PROF_" + name)) for name in ("SYRK", "COV", "PANEL", "SOLVE", "PREDICT")}
    if how == "none":
        ctx.profile_enable(False)
    elif syrk_only:
        ctx.profile_enable(True, only=g._lib.PROF_SYRK)
    else:
        ctx.profile_enable(True)
    barrier()
    wall["fit"] = wall["predict"] = 0.0
    t0 = time.perf_counter()
    for i in range(steps):
        mu, s2 = step(warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    syrk_bytes = ctx.profile_get_bytes(g._lib.PROF_SYRK)
    prof = {"SYRK": ctx.profile_get(g._lib.PROF_SYRK)}
    if sharded:  # per-step phases of the blocked driver (HIP events on this rank; include/gpmi.h GPMI_PROF_STEP_*)
        prof["phases"] = {name: ctx.profile_get(cls) for name, cls in g._lib.STEP_PHASES.items()}
    ctx.profile_enable(False)
    prof["stage"] = stage
    prof["stage_steps"] = warmup
    assert np.all(np.isfinite(mu)) and np.all(np.isfinite(s2)) and math.isfinite(gp.mll)
    return {"elapsed": elapsed, "prof": prof, "syrk_bytes": syrk_bytes, "t_build": t_build, "mll": gp.mll, "ll": ll,
            "fit_ms": 1e3 * wall["fit"] / steps, "predict_ms": 1e3 * wall["predict"] / steps,
            "base_mll": base_mll, "base_mu": np.asarray(base_mu, dtype=np.float64), "base_s2": np.asarray(base_s2, dtype=np.float64),
            "base_alpha": base_alpha}


def phases_object(res, steps):
    """per_step_ms: what one block step of the sharded factorisation spends in each phase ON THIS RANK (HIP events on the stream the
    phase runs on; the phases overlap by design — chain + broadcast under U2a, gather under U2b — so they do not add up to the step)."""
    ph = res["prof"].get("phases")
    if not ph:
        return None
    out = {}
    for name, (cnt, ms, _w) in ph.items():
        out[name] = {"ms_per_block_step": (ms / cnt) if cnt else None, "ms_per_fit": ms / max(steps, 1), "block_steps_per_fit": cnt / max(steps, 1)}
    out["note"] = ("rank 0; chain = factor + explicit inverse of the next diagonal block (its owner only), broadcast = that inverse, solve = next "
                   "panel X LW', gather = all-gather of the solved panel; U2a hides chain + broadcast, U2b hides the gather")
    return out


_PEAK_MEASURED = {}


def roofline_object(args, res, n, d, p, dtype, steps, ctx=None):
    peak = FP64_MFMA_PEAK_TFLOPS if dtype == "f64" else FP32_MFMA_PEAK_TFLOPS
    n_syrk, ms_syrk, fl_syrk = res["prof"]["SYRK"]
    achieved = (fl_syrk / max(ms_syrk, 1e-9)) * 1e-9  # flop/ms -> TFLOP/s
    es = 8 if dtype == "f64" else 4
    extra = {}
    if ctx is not None:
        try:  # the instruction-rate ceiling of THIS chip in THIS run: every SIMD issuing back-to-back MFMAs (gpmi_mfma_peak)
            bits = 64 if dtype == "f64" else 32
            samples = _PEAK_MEASURED.setdefault(bits, [])
            samples.extend(ctx.mfma_peak(bits) for _ in range(3))
            best = max(samples)
            extra = {"peak_measured": best, "frac_of_measured": achieved / best, "peak_measured_samples": [round(v, 2) for v in samples],
                     "peak_measured_note": "gpmi_mfma_peak of this run, sampled before the first fit and after the timed steps (best sample): the update "
                                           "kernel's OWN K loop with everything but its MFMAs compiled out (update256_kernel<T, ABL = 15>: no operand DMA, no "
                                           "fragment reads, no slab barrier, no epilogue; 4096 tiles of 256 x 128 x 2048 through the same per-XCD queues, "
                                           "random operands) - the rate the product kernel would reach if memory, LDS and barriers were free, at the clock "
                                           "the chip sustains under that stream.  `frac` stays against the spec peak (256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)"}
        except Exception:  # noqa: BLE001
            extra = {}
    return {
        **extra,
        "kernel": "Cholesky trailing update, K = super-panel width, v_mfma_f64_16x16x4: update256_kernel<T> (256x128 tiles: launches of >= 1024 tiles "
                  "in factorisations of >= 12288 rows) + gemm_nt_kernel<T, 0, 4> (128x128 tiles, the others); all launches of the class are averaged",
        "bound": "mfma",
        "achieved": achieved,
        "peak": peak,
        "unit": "TFLOP/s",
        "frac": achieved / peak,
        **_pmc_traffic(args, n, d, p),
        "launches": n_syrk,
        "avg_launch_ms": ms_syrk / max(n_syrk, 1),
        "algorithmic_flops_per_launch": fl_syrk / max(n_syrk, 1),
        # every trailing entry read + written once per launch, the factored panel read once (K = 256 ... 1024 per launch)
        "algorithmic_bytes_per_launch": res["syrk_bytes"] / max(n_syrk, 1),
    }


def stage_object(res, steps):
    pr = res["prof"].get("stage")
    steps = res["prof"].get("stage_steps", 0)
    if not pr or steps <= 0:
        return {"note": "no warm-up steps: stage times not taken"}
    return {
        "cov": pr["COV"][1] / steps,
        "cov_GBps": (pr["COV"][2] / max(pr["COV"][1], 1e-9)) * 1e-6,
        "chol_trailing_update": pr["SYRK"][1] / steps,
        "panel_potf2_trsm_update": pr["PANEL"][1] / steps,
        "alpha_solve_mll": pr["SOLVE"][1] / steps,
        "predict": pr["PREDICT"][1] / steps,
        "note": "per-class sums of HIP-event intervals over the WARM-UP steps; the panel chain runs on a side stream UNDER the trailing "
                "update (look-ahead), so the classes overlap and do not add up to ms_per_step",
    }


def parity_object(res, cpu, n, dtype):
    """Full-size parity of THIS line's workload: the device fit at the base hyper-parameters (the constructor's fit and a
    predict_f on it, run_workload) against the cpu_baseline oracle run at the same N (same inputs, same parameters).
    north_star's bar: rtol 1e-5 (fp64) / 1e-2 (fp32) on log-mll, posterior mean and variance."""
    tol = 1e-5 if dtype == "f64" else 1e-2
    mu_o, s2_o = cpu.pop("_mu", None), cpu.pop("_s2", None)
    if mu_o is None or cpu.get("n_measured") != n:
        return {"checked": False, "reason": f"the CPU oracle ran at N={cpu.get('n_measured')}, not at the bench size N={n}"}
    mu, s2 = res["base_mu"], res["base_s2"]
    mll_rel = abs(res["base_mll"] - cpu["mll"]) / abs(cpu["mll"])
    mu_rel = float(np.abs(mu - mu_o).max() / np.abs(mu_o).max())
    s2_rel = float(np.abs(s2 - s2_o).max() / np.abs(s2_o).max())
    elem_ok = bool(np.allclose(mu, mu_o, rtol=tol, atol=tol * np.abs(mu_o).max()) and
                   np.allclose(s2, s2_o, rtol=tol, atol=tol * np.abs(s2_o).max()))
    return {
        "checked": True,
        "against": "cpu_baseline (oracle/gp_oracle + oracle/cov_oracle.c + LAPACK) at the same N, inputs and hyper-parameters",
        "n": n,
        "p": int(mu.shape[0]),
        "tol": tol,
        "mll_device": res["base_mll"],
        "mll_oracle": cpu["mll"],
        "mll_rel_err": mll_rel,
        "mu_max_err_over_max_abs": mu_rel,
        "var_max_err_over_max_abs": s2_rel,
        "ok": bool(mll_rel <= tol and mu_rel <= tol and s2_rel <= tol and elem_ok),
    }


C3_SPEC_NOTE = "Sum(Sum(SEArd, Mat52Iso), Noise)"


def sparse_probe_parity(res, n, d, p, tol, spec=None):
    """`parity` of a fit too large for a host factorisation (C4: N = 200 000): the device's alpha at the base hyper-parameters must
    satisfy (K + s2 I) alpha = y on 512 random rows of K REBUILT by the fp64 oracle, and mu = K*' alpha on 64 test points rebuilt the
    same way (the checks of tests/test_gpu_fullsize.py::test_c4_fp32_n200000..., inside the driver's own run)."""
    from oracle import gp_oracle as G

    x, y, xpred = synthetic_inputs(n, d, p)
    spec = spec or ("se_ard", _ll(d), 0.0)
    nv = math.exp(2.0 * math.log(0.1))
    a = res["base_alpha"]
    rows = np.sort(np.random.default_rng(7).choice(n, 512, replace=False))
    r = G.cov(spec, x[:, rows], x) @ a + nv * a[rows] - y[rows]
    resid = float(np.abs(r).max() / np.abs(y).max())
    mu_o = G.cov(spec, xpred[:, :64], x) @ a
    mu_err = float(np.abs(res["base_mu"][:64] - mu_o).max() / max(np.abs(mu_o).max(), 1e-300))
    return {"checked": True, "what": "solve residual |(K + s2 I) alpha - y| / max|y| on 512 oracle-rebuilt rows; mu = K*' alpha on 64 oracle-rebuilt "
                                     "test points", "residual_over_max_abs_y": resid, "mu_max_err_over_max_abs": mu_err, "tol": tol,
            "ok": bool(resid <= tol and mu_err <= tol)}


def run_c3(g, ctx, steps=3):
    """BASELINE configs[2]: N = 50 000, d = 8, (SEArd + Mat52Iso) + Noise, fp64 — the composite-kernel cov! path."""
    import gc

    gc.collect()
    n, d, p = 50000, 8, 1024
    x, y, xpred = synthetic_inputs(n, d, p)
    spec = ("sum", ("sum", ("se_ard", _ll(d), 0.0), ("mat52_iso", math.log(0.7), math.log(0.5))), ("noise", math.log(0.05)))
    gp = g.GP(x, y, g.MeanZero(), g.from_spec(spec), math.log(0.1), ctx=ctx)
    base = np.asarray(gp.get_params())
    mll0 = gp.mll
    mu0, _ = gp.predict_f(xpred)
    par = sparse_probe_parity({"base_alpha": np.asarray(gp.alpha, dtype=np.float64), "base_mu": np.asarray(mu0, dtype=np.float64)}, n, d, p, 1e-5,
                              spec=spec)
    ctx.profile_enable(True, only=g._lib.PROF_COV)
    t0 = time.perf_counter()
    for i in range(steps):
        gp.set_params(base + 1e-3 * (i + 1) * np.where(np.arange(len(base)) == 0, 0.0, 1.0))
        gp.update_mll()
        mu, s2 = gp.predict_f(xpred)
    el = time.perf_counter() - t0
    cov = ctx.profile_get(g._lib.PROF_COV)
    ctx.profile_enable(False)
    assert np.all(np.isfinite(mu)) and np.all(s2 >= 0) and math.isfinite(gp.mll)
    return {
        "workload": f"N={n}, d={d}, {C3_SPEC_NOTE} + MeanZero, f64, P={p} (BASELINE.json configs[2]), {steps} steps",
        "ms_per_step": 1e3 * el / steps,
        "fits_per_sec": steps / el,
        "mll_base_params": mll0,
        "cov_ms_per_step": cov[1] / steps,
        "cov_GBps": (cov[2] / max(cov[1], 1e-9)) * 1e-6,
        "parity": par,
    }


def directional_parity(gp, h=1e-3, tol=1e-4):
    """`parity` of a gradient at full size: the analytic directional derivative dmll . v against the central difference of the
    device mll (itself oracle-checked in `parity` / the -m gpu tests) along a fixed direction v over ALL parameters
    [logNoise; kernel...] — two extra fits (test/kernels.jl:148-164 checks dtarget against a finite difference the same way)."""
    base = np.asarray(gp.get_params(), dtype=np.float64)
    grad = np.asarray(gp.dmll, dtype=np.float64).copy()
    rng = np.random.default_rng(123)
    v = rng.uniform(0.5, 1.0, size=base.shape) * np.where(rng.uniform(size=base.shape) < 0.5, -1.0, 1.0)
    f = []
    for sgn in (1.0, -1.0):
        gp.set_params(base + sgn * h * v)
        gp.update_mll()
        f.append(gp.mll)
    gp.set_params(base)
    fd = (f[0] - f[1]) / (2.0 * h)
    an = float(grad @ v)
    rel = abs(an - fd) / max(abs(fd), 1e-300)
    return {"checked": True, "what": "dmll . v vs central difference of the device mll along a fixed direction over all parameters",
            "h": h, "analytic": an, "central_difference": fd, "rel_err": rel, "tol": tol, "ok": bool(rel <= tol)}


def run_grad(g, ctx, n=50000, d=8):
    """update_dmll! (src/GPE.jl:298-324) at the bench size: K^-1 via the whitened identity + the fused dK/dtheta trace pass."""
    import gc

    gc.collect()
    x, y, _ = synthetic_inputs(n, d, 8)
    gp = g.GP(x, y, g.MeanZero(), g.SEArd(_ll(d), 0.0), math.log(0.1), ctx=ctx)
    gp.update_dmll()                      # first call allocates the two extra N x N buffers
    t0 = time.perf_counter()
    gp.update_dmll()
    el = time.perf_counter() - t0
    fl = 2.0 * float(n) ** 3 / 3.0        # L^-T rows (n^3/3) + K^-1 = L^-T L^-1 lower (n^3/3)
    return {
        "workload": f"update_dmll! at N={n}, d={d}, SEArd, f64 (after update_mll!; {len(gp.dmll)} parameters)",
        "s_per_call": el,
        "TFLOPs_on_2n3_over_3": fl / el * 1e-12,
        "frac_of_fp64_matrix_peak": fl / el * 1e-12 / FP64_MFMA_PEAK_TFLOPS,
        "mll": gp.mll,
        "dmll_inf_norm": float(np.abs(gp.dmll).max()),
        "parity": directional_parity(gp),
    }


def run_c5(g, ctx, n=1000000, m=4096, d=8):
    """BASELINE configs[4]: FITC, N = 1e6, M = 4096 inducing points, SEArd, fp64, one GPU: update_mll!, predict_f, update_dmll!."""
    import gc

    gc.collect()
    rng = np.random.default_rng(20240501)
    x = rng.uniform(size=(d, n))
    xu = rng.uniform(size=(d, m))
    y = np.sin(2.0 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
    xs = rng.uniform(size=(d, 1024))
    t0 = time.perf_counter()
    gp = g.FITC(x, xu, y, g.MeanZero(), g.SEArd(_ll(d), 0.0), math.log(0.1), ctx=ctx)
    t_first = time.perf_counter() - t0
    mll0 = gp.mll
    gp.set_params([v + 0.01 for v in gp.get_params()])
    t0 = time.perf_counter()
    gp.update_mll()
    t_fit = time.perf_counter() - t0
    t0 = time.perf_counter()
    mu, var = gp.predict_f(xs)
    t_pred = time.perf_counter() - t0
    gp.update_dmll()                      # first call allocates two more n x m buffers
    t0 = time.perf_counter()
    gp.update_dmll()
    t_grad = time.perf_counter() - t0
    assert np.all(np.isfinite(mu)) and np.all(var >= 0) and math.isfinite(gp.mll)
    fl = 2.0 * n * float(m) * m           # W = Kfu Luu^-T (n m^2) + U'U'^T (n m^2)
    mll_timed, dinf = gp.mll, float(np.abs(gp.dmll).max())
    par = directional_parity(gp)
    return {
        "workload": f"FITC N={n}, M={m}, d={d}, SEArd + MeanZero, f64 (BASELINE.json configs[4])",
        "first_fit_incl_alloc_upload_s": t_first,
        "update_mll_s": t_fit,
        "update_mll_TFLOPs_on_2nm2": fl / t_fit * 1e-12,
        "update_mll_frac_of_fp64_matrix_peak": fl / t_fit * 1e-12 / FP64_MFMA_PEAK_TFLOPS,
        "predict_f_1024_ms": 1e3 * t_pred,
        "update_dmll_s": t_grad,
        "mll_base_params": mll0,
        "mll": mll_timed,
        "dmll_inf_norm": dinf,
        "parity": par,
    }


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=50000)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--p", type=int, default=1024)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--cpu-budget-s", type=float, default=240.0,
                    help="the CPU baseline is measured at the bench size when a probe estimates it under this many seconds")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the c2 / c4 / c3 / grad / c5 secondary objects")
    ap.add_argument("--secondary", default="c2,c4,c3,grad,c5", help="comma-separated subset of the secondary objects to run")
    ap.add_argument("--dry-run-one-gpu", action="store_true",
                    help="REHEARSAL of the multi-GPU line on ONE GPU: the N ranks are processes on device 0 (two ranks: its two CU partitions), "
                         "joined by torch.distributed gloo on libgpmi's device buffers instead of RCCL; the line says so in config.parallelism")
    ap.add_argument("--c4-n", type=int, default=200000, help="size of the c4 secondary object (the rehearsal shrinks it: gloo moves panels through the host)")
    ap.add_argument("--mode", default=None, choices=["replicas", "sharded"],
                    help="N>1: ONE fit row-block sharded over the GPUs with the RCCL panel exchange (default, strong scaling) "
                         "or independent fits per GPU (replicas, weak scaling); N=1: sharded runs the sharded code path on one GPU")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        # no launcher: start one process per GPU ourselves (the contract's torchrun line)
        import torch

        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if args.dry_run_one_gpu:
            have = args.gpus if have >= 1 else 0
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: this box has {have} GPU(s) visible; refusing to run fewer ranks "
                             "than asked (no silent downgrade)")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "--", os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    dist = torch = None
    if world > 1:
        # one rank per GPU: torch.distributed is the launcher's rendezvous, barrier and max-over-ranks (imported BEFORE libgpmi so that
        # the two share one HIP runtime).  The single-GPU process does not load torch at all: libgpmi then runs on ROCm's own HIP
        # runtime instead of the older one bundled in the torch wheel (measured: N = 20 000 68.5 instead of 73.4 ms per step), and the
        # bracket around the timed region is gpmi_ctx_synchronize (hipDeviceSynchronize) in place of torch.cuda.synchronize().
        import torch
        import torch.distributed as dist

        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a MI355X: no GPU visible (there is no CPU fallback)")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime

        # a collective that never completes becomes an error after 10 minutes instead of a hang until the caller's limit
        if args.dry_run_one_gpu:
            torch.cuda.set_device(0)
            dist.init_process_group(backend="gloo", timeout=datetime.timedelta(minutes=10))
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(minutes=10))

    import gpmi355x as g

    mode = args.mode or ("sharded" if world > 1 else "single")
    sharded = mode == "sharded"
    n, d, p = args.n, args.d, args.p
    try:
        if args.dry_run_one_gpu and world > 1:
            ctx = g.Context(256 * (1 + rank) if world == 2 else 0)   # two ranks: the two CU partitions of device 0 (include/gpmi.h)
        else:
            ctx = g.Context.default(local_rank)
    except g._lib.DeviceError as e:
        raise SystemExit(f"bench.py needs a MI355X: {e} (there is no CPU fallback)")

    def barrier():
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        ctx.synchronize()

    _comm = []

    _comm_kind = []

    def make_comm():
        """ONE communicator per process, verified before the first fit: libgpmi's own RCCL communicator (the unique id travels
        through the launcher's torch.distributed group), or — GPMI_DIST_COMM=torch, and as the fallback when every rank agrees that
        the native one could not be opened or failed its self-test — torch.distributed (backend nccl = RCCL) behind the callbacks."""
        if dist is None:
            return None
        if not _comm:
            from gpmi355x import dist as gd

            c = None
            if args.dry_run_one_gpu:
                c = gd.TorchDistComm(device=0)
                c.selftest(ctx)
                _comm_kind.append("REHEARSAL on one GPU: torch.distributed gloo on device buffers behind gpmi_comm_callbacks (not RCCL)")
            elif os.environ.get("GPMI_DIST_COMM", "rccl") != "torch":
                ok = 1
                try:
                    c = gd.rccl_comm(ctx)
                    c.selftest(ctx)
                except Exception as e:  # noqa: BLE001
                    ok = 0
                    sys.stderr.write(f"bench.py rank {rank}: native RCCL communicator unavailable ({e!r}); asking the other ranks\n")
                t = torch.tensor([ok], dtype=torch.int32, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MIN)  # (nccl only: the rehearsal takes the branch above)
                if int(t.item()) == 0:  # somebody failed: everybody falls back (the collectives must match on every rank)
                    if c is not None:
                        c.close()
                    c = None
                else:
                    _comm_kind.append("libgpmi RCCL (gpmi_comm_create_rccl)")
            if c is None:
                c = gd.TorchDistComm(device=local_rank)
                c.selftest(ctx)
                _comm_kind.append("torch.distributed nccl behind gpmi_comm_callbacks")
            _comm.append(c)
        return _comm[0]

    def max_over_ranks(v):
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device="cpu" if args.dry_run_one_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    try:  # the instruction-rate ceiling on the idle, cool chip (sampled again after the timed steps: roofline_object)
        _PEAK_MEASURED.setdefault(64 if args.dtype == "f64" else 32, []).extend(ctx.mfma_peak(64 if args.dtype == "f64" else 32) for _ in range(3))
    except Exception:  # noqa: BLE001
        pass
    res = run_workload(g, ctx, n, d, p, args.dtype, args.steps, args.warmup, barrier, comm=make_comm() if sharded else None, sharded=sharded)
    elapsed = max_over_ranks(res["elapsed"])

    out = None
    if rank == 0:
        nfits = args.steps if (sharded or world == 1) else world * args.steps
        if world == 1:
            par = "single GPU" + (" (sharded code path, one rank)" if sharded else "")
        elif sharded:
            par = (f"ONE fit row-block sharded over {world} GPUs (block-cyclic super-panel blocks of 1024 rows; per block an RCCL broadcast of the "
                   "diagonal block's inverse and an all-gather of the solved panel, both under the trailing update: csrc/blocked.cpp)")
            if args.dry_run_one_gpu:
                par = (f"REHEARSAL (--dry-run-one-gpu): ONE fit row-block sharded over {world} PROCESSES that share ONE MI355X "
                       + ("(its two CU partitions) " if world == 2 else "") + "joined by gloo through the host — NOT a multi-GPU measurement; "
                       "it exists to prove that the multi-GPU line, its parity and its secondary objects come out")
        else:
            par = f"{world} independent fits, one per GPU (no data-path collective)"
        out = {
            "metric": "GP fits/sec (update_mll!+predict_f)",
            "value": nfits / elapsed,
            "unit": "GP fits/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "fit_only_ms_per_step": res["fit_ms"],          # update_mll! alone (SURVEY 8d asks for fit-only and fit+predict), host wall clock
            "predict_only_ms_per_step": res["predict_ms"],
            "higher_is_better": True,
            "scaling": "weak" if (world > 1 and not sharded) else "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"N={n}, d={d}, SEArd + MeanZero, {args.dtype}, P={p} test points, full_cov=false "
                            "(the configuration BASELINE.json's north-star target is quoted on)",
                "parallelism": par,
                "mll": res["mll"],
                **({"communicator": _comm_kind[0]} if _comm_kind else {}),
            },
            "roofline": roofline_object(args, res, n, d, p, args.dtype, args.steps, ctx),
            "stage_ms_per_step": stage_object(res, args.steps),
            "first_fit_incl_upload_s": res["t_build"],
            **({"per_step_ms": phases_object(res, args.steps)} if sharded else {}),
        }

    want = set(args.secondary.split(","))

    def sec_c2(sec):
        c2 = run_workload(g, ctx, 20000, 8, 1024, "f64", 5, 2, barrier, comm=make_comm() if sharded else None, sharded=sharded)
        c2_el = max_over_ranks(c2["elapsed"])
        sec["c2"] = {
            "workload": "N=20000, d=8, SEArd + MeanZero, f64, P=1024 (BASELINE.json configs[1]), 5 steps after 2 warm-ups",
            "ms_per_step": 1e3 * c2_el / 5,
            "fit_only_ms_per_step": c2["fit_ms"],
            "predict_only_ms_per_step": c2["predict_ms"],
            "fits_per_sec": 5 / c2_el,
            "roofline_frac": roofline_object(args, c2, 20000, 8, 1024, "f64", 5)["frac"],
            "stage_ms_per_step": {k: v for k, v in stage_object(c2, 5).items() if k != "note"},
            **({"per_step_ms": phases_object(c2, 5)} if (sharded and rank == 0) else {}),
        }

    def sec_c4(sec):
        # north_star's multi-GPU size as ONE fit: on one GPU (160 GB of fp32 factor fit the 288 GB) or sharded
        n4 = args.c4_n
        c4 = run_workload(g, ctx, n4, 16, 1024, "f32", 1, 0, barrier, comm=make_comm() if sharded else None, sharded=sharded)
        c4_el = max_over_ranks(c4["elapsed"])
        sec["c4_sharded" if (sharded and world > 1) else "c4_single_gpu"] = {
            "workload": f"N={n4}, d=16, SEArd + MeanZero, f32, P=1024: ONE fit+predict on {world} GPU(s), 1 step after the "
                        "constructor's fit (" + ("BASELINE.json configs[3]'s size)" if n4 == 200000 else "a SHRUNK stand-in for BASELINE.json configs[3]: --c4-n)"),
            "s_per_step": c4_el,
            "fits_per_sec": 1.0 / c4_el,
            "chol_equiv_TFLOPs": (float(n4) ** 3 / 3.0) / c4_el * 1e-12,
            "mll": c4["mll"],
            **({"per_step_ms": phases_object(c4, 1)} if (sharded and rank == 0) else {}),
            **({"parity": sparse_probe_parity(c4, n4, 16, 1024, 1e-2)} if rank == 0 else {}),
        }

    def secondaries(sec, order=("c2", "c4")):
        """c2 / c4 / c3 / grad / c5 objects (none of them is `value`)."""
        for key in order:
            if key not in want:
                continue
            try:
                (sec_c2 if key == "c2" else sec_c4)(sec)
            except Exception as e:  # noqa: BLE001
                sec[key if key == "c2" else "c4_error"] = {"error": repr(e)[:300]} if key == "c2" else repr(e)[:300]
        if world == 1 and not sharded:
            for key, fn in (("c3", run_c3), ("grad", run_grad), ("c5", run_c5)):
                if key not in want:
                    continue
                try:
                    sec[key] = fn(g, ctx)
                except Exception as e:  # noqa: BLE001
                    sec[key] = {"error": repr(e)[:300]}

    exit_code = 0
    if world == 1:
        if not args.no_secondary:
            sec = {}
            secondaries(sec)
            out.update(sec)
        parity_failed = False
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(n, d, p, res["ll"], args.cpu_budget_s)
            out["parity"] = parity_object(res, cpu, n, args.dtype)
            cpu.pop("_mu", None)
            cpu.pop("_s2", None)
            out["cpu_baseline"] = cpu
            parity_failed = out["parity"].get("checked", False) and not out["parity"]["ok"]
        print(json.dumps(out), flush=True)
        if parity_failed:
            sys.stderr.write("bench.py: PARITY FAILED against the CPU oracle at the bench size: " + json.dumps(out["parity"]) + "\n")
            sys.exit(3)
    else:
        # ONE JSON line.  What a first multi-GPU run needs to be informative rides IN it: `parity` of the sharded fit (solve residual on
        # oracle-rebuilt rows, rank 0, ~2 s), `per_step_ms` (already measured in the timed region) and `c4_sharded` — north_star's
        # N = 200 000 fp32 configuration on the same ranks, with its own parity.  A watchdog on EVERY rank prints the line with whatever is
        # there when the extras have not finished in time (all ranks leave together, so nobody waits in a collective for a rank that left).
        import copy
        import threading

        lock = threading.Lock()  # guards `out` (the main thread adds to it, the watchdog prints it) and `printed`
        printed = [False]
        done = threading.Event()

        def put(**kv):
            with lock:
                out.update(kv)

        def persist(tag):
            """the line as it stands, on disk: a secondary that takes the process down (a device fault, an OOM kill, torchrun tearing the
            ranks down) must not take the primary measurement with it"""
            if rank != 0:
                return
            try:
                with lock:
                    snap = copy.deepcopy(out)
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                with open(os.path.join(ROOT, "gpurun_out", f"bench_gpus{world}{tag}.json"), "w") as fh:
                    json.dump(snap, fh)
            except OSError:
                pass

        def emit(note=None):
            with lock:
                if printed[0]:
                    return
                if rank == 0:
                    snap = copy.deepcopy(out)
                    if note:
                        snap["secondary_note"] = note
                    print(json.dumps(snap), flush=True)
                printed[0] = True  # only after the line is out: a failed dump leaves the other thread free to try again

        budget_s = float(os.environ.get("GPMI_BENCH_SECONDARY_S", "420"))

        def watchdog():
            if not done.wait(budget_s):
                emit(f"the secondary objects did not finish within {budget_s:.0f} s (watchdog): the line was printed without them")
                os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()
        persist("_primary")  # the timed region's result, before anything else runs
        if rank == 0:
            put(per_step_ms=phases_object(res, args.steps))
            try:
                put(parity=sparse_probe_parity(res, n, d, p, 1e-5 if args.dtype == "f64" else 1e-2))
            except Exception as e:  # noqa: BLE001
                put(parity={"checked": False, "error": repr(e)[:300]})
            persist("_primary")
        if not args.no_secondary:
            sec = {}
            try:
                secondaries(sec, order=("c4", "c2"))
            except Exception as e:  # noqa: BLE001  (a collective that failed on this rank: say so, print what there is)
                sec["secondary_error"] = repr(e)[:300]
            if rank == 0:
                put(**sec)
        done.set()
        emit()
        persist("")
        with lock:
            par = out.get("parity") if isinstance(out, dict) else None  # (`out` exists on rank 0 only)
        if rank == 0 and isinstance(par, dict) and par.get("checked") and not par.get("ok"):
            sys.stderr.write("bench.py: PARITY FAILED (sharded fit, solve residual on oracle-rebuilt rows): " + json.dumps(par) + "\n")
            exit_code = 3

    if dist is not None:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
    if world > 1:
        sys.stdout.flush()
        os._exit(exit_code)  # skip interpreter teardown of RCCL / HIP objects in multi-process runs


if __name__ == "__main__":
    main()
