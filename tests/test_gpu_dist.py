"""Blocked / sharded path on the GPU: the orchestration below the C ABI (gpmi_gp_create_blocked) on the real HIP kernels,
  (a) as a single rank — plain rows and packed stripes — including the gradient and optimize,
  (b) as G virtual ranks (threads, each with its own gpmi context) sharing the one GPU through an in-process communicator
      built on gpmi_comm_callbacks,
  (c) as TWO PROCESSES on the one GPU under real torch.distributed collectives (gloo on device buffers, through the product's
      TorchDistComm) at N = 12 288 — stream ordering between asynchronous collectives and gpmi kernels across processes,
  (d) RCCL itself in a group of one: libgpmi's own RCCL communicator (gpmi_comm_create_rccl) and torch's "nccl" backend behind
      the callbacks, each through gpmi_comm_selftest,
  (e) (round 4) as an in-process device group over the two CU PARTITIONS of the GPU (device ids 256 / 512: disjoint halves of
      every XCD — two logical devices that run side by side), incl. the injected-latency measurement of what the look-ahead
      pipeline hides, and the AbstractPDMat surface (solve / whiten / inv_diag / cholfactors) on every blocked configuration.
Together with tests/test_blocked_cpu.py (the same orchestration source on a host stand-in: thread ranks and gloo process groups
of 2 and 3) this covers the multi-GPU path as far as one GPU allows; what cannot run here is RCCL between >= 2 devices
(tools/multidev_check.py is the functional check for a box that shows more than one)."""
import math
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import gpmi355x as g
from gpmi355x import dist as gd
from dist_helpers import LocalThreadComm
from oracle import gp_oracle as G

pytestmark = pytest.mark.gpu

SPEC = ("sum", ("se_ard", [-0.5, -0.3, -0.6, -0.2], 0.2), ("mat52_iso", -0.4, -0.5))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(n, p=45, seed=3):
    rng = np.random.default_rng(seed)
    x = rng.uniform(size=(4, n))
    y = np.sin(3 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
    xs = rng.uniform(size=(4, p))
    return x, y, xs


def _check(gp, x, y, xs, ln, mspec, rtol_mll=1e-10, grad=False):
    ref = G.update_mll(SPEC, x, y, ln, mspec)
    assert abs(gp.mll - ref["mll"]) <= rtol_mll * abs(ref["mll"]), (gp.mll, ref["mll"])
    np.testing.assert_allclose(gp.alpha, ref["alpha"], rtol=1e-7, atol=1e-8 * np.abs(ref["alpha"]).max())
    assert abs(gp.logdet - ref["logdet"]) <= 1e-10 * abs(ref["logdet"])
    np.testing.assert_allclose(gp.cK.factor_diag(), np.diag(ref["U"]), rtol=1e-9)
    mu, s2 = gp.predict_f(xs)
    mu_o, s2_o = G.predict_f(SPEC, x, ref, xs, mspec)
    np.testing.assert_allclose(mu, mu_o, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(s2, s2_o, rtol=1e-6, atol=1e-10)
    mu_f, S = gp.predict_f(xs, full_cov=True)                      # the other branch of predict_f (GP.jl:80-84)
    _, S_o = G.predict_f(SPEC, x, ref, xs, mspec, full_cov=True)
    np.testing.assert_allclose(mu_f, mu_o, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(S, S_o, rtol=1e-6, atol=1e-9)
    if grad:
        gp.update_dmll()
        d = G.update_dmll(SPEC, x, y, ln, mspec, fit=ref)["dmll"]
        np.testing.assert_allclose(gp.dmll, d, rtol=1e-6, atol=1e-8 * np.abs(d).max())
        _check_pdmat(gp, ref, y, mspec)


def _check_pdmat(gp, ref, y, mspec):
    """the AbstractPDMat surface on a blocked handle (VERDICT r3 missing 2): `\\`, whiten!, diag(inv(cK)) -> predict_LOO, cholfactors,
    and update_mll!(kern = false, noise = false) = `cK \\ (y - mu)` on the kept factor (GPE.jl:203-211), against LAPACK on the oracle's U"""
    import scipy.linalg as sla

    U = np.triu(ref["U"])
    n = U.shape[0]
    B = np.random.default_rng(3).standard_normal((n, 3))
    np.testing.assert_allclose(gp.cK.cholfactors(), U, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(gp.cK.whiten(B), sla.solve_triangular(U, B, trans="T", lower=False), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(gp.cK.solve(B), sla.cho_solve((U, False), B), rtol=1e-6, atol=1e-8 * np.abs(B).max())
    Kinv_diag = np.diag(sla.cho_solve((U, False), np.eye(n)))
    np.testing.assert_allclose(gp.cK.inv_diag(), Kinv_diag, rtol=1e-7)
    mu_loo, s2_loo = gp.predict_LOO()                                # crossvalidation.jl:8-13
    np.testing.assert_allclose(s2_loo, 1.0 / Kinv_diag, rtol=1e-7)
    np.testing.assert_allclose(mu_loo, y - ref["alpha"] / Kinv_diag, rtol=1e-6, atol=1e-8)
    mll0, a0 = gp.mll, np.array(gp.alpha)
    gp.update_mll(kern=False, noise=False)                           # the factor is kept; alpha and mll are re-derived from it
    assert abs(gp.mll - mll0) <= 1e-10 * abs(mll0)
    np.testing.assert_allclose(gp.alpha, a0, rtol=1e-7, atol=1e-9 * np.abs(a0).max())
    #  ... and after a CHANGE of the mean the device copy of alpha must follow: predict_f / update_dmll! read it
    if gp.mean.num_params() > 0:
        p0 = list(gp.mean.get_params())
        gp.mean.set_params([v + 0.3 for v in p0])
        gp.update_mll(kern=False, noise=False)
        xs = np.ascontiguousarray(gp.x[:, :9], dtype=np.float64) + 0.01
        mu_s, _ = gp.predict_f(xs)
        gp.update_dmll()
        d_s = np.array(gp.dmll)
        gp.update_mll()
        mu_r, _ = gp.predict_f(xs)
        gp.update_dmll()
        np.testing.assert_allclose(mu_s, mu_r, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(d_s, gp.dmll, rtol=1e-6, atol=1e-8 * np.abs(gp.dmll).max())
        gp.mean.set_params(p0)
        gp.update_mll()
    draws = gp.rand(np.ascontiguousarray(gp.x[:, :7], dtype=np.float64), n=3, rng=np.random.default_rng(0))   # GP.jl:120-146 on a blocked model
    assert draws.shape == (7, 3) and np.all(np.isfinite(draws))


@pytest.mark.parametrize("n,block", [(300, None), (1000, None), (1793, None), (5000, None), (3000, 1024), (4500, 2048)])
def test_single_rank_device_ops(n, block):
    """block None: 256-row blocks below 4096 points, 512 from there; every step runs the look-ahead pipeline (next diagonal
    block factored and inverted on the chain stream under the update)."""
    x, y, xs = _problem(n)
    ln = math.log(0.1)
    gp = gd.ShardedGPE(x, y, g.MeanConst(0.1), g.from_spec(SPEC), ln, block=block)
    assert gp.WD == (block or (512 if n >= 4096 else 256))
    _check(gp, x, y, xs, ln, ("const", 0.1), grad=True)


@pytest.mark.parametrize("n,block,per", [(2300, 256, 2), (5000, 512, 3), (9000, 1024, 2)])
def test_single_rank_packed_stripes(n, block, per):
    """SURVEY §8f-3 on the device: stripes that stop at their own diagonal; every update is one launch per stripe."""
    x, y, xs = _problem(n)
    ln = math.log(0.1)
    gp = g.GP(x, y, g.MeanConst(0.1), g.from_spec(SPEC), ln, packed=True, block=block, stripe_blocks=per)
    assert gp.cK.nstripes > 1 and gp.cK.factor_bytes < 0.8 * 8 * (gp.nblk * gp.WD) ** 2
    _check(gp, x, y, xs, ln, ("const", 0.1), grad=True)


def test_fp32_blocked_handle_vs_fp64_oracle():
    x, y, xs = _problem(5000)
    ln = math.log(0.1)
    gp = g.GP(x, y, g.MeanZero(), g.from_spec(SPEC), ln, packed=True, block=512, stripe_blocks=3, dtype=np.float32)
    ref = G.update_mll(SPEC, x, y, ln)
    assert abs(gp.mll - ref["mll"]) <= 1e-2 * abs(ref["mll"])
    mu, s2 = gp.predict_f(xs)
    mu_o, s2_o = G.predict_f(SPEC, x, ref, xs)
    np.testing.assert_allclose(mu, mu_o, rtol=1e-2, atol=1e-2 * np.abs(mu_o).max())
    np.testing.assert_allclose(s2, s2_o, rtol=1e-2, atol=1e-2 * np.abs(s2_o).max())


def test_optimize_on_a_packed_model_reaches_the_dense_optimum():
    """optimize! (src/optimize.jl:19-37) on a blocked handle: target and gradient from gpmi_fit / gpmi_grad of the blocked driver;
    same optimum as the dense device path from the same start."""
    x, y, _ = _problem(1500)
    k1 = g.SEArd([0.0, 0.0, 0.0, 0.0], 0.0)
    k2 = g.SEArd([0.0, 0.0, 0.0, 0.0], 0.0)
    dense = g.GP(x, y, g.MeanZero(), k1, -1.0)
    packed = g.GP(x, y, g.MeanZero(), k2, -1.0, packed=True, block=256, stripe_blocks=2)
    assert packed.mll == pytest.approx(dense.mll, rel=1e-10)
    rd = g.optimize(dense, options={"maxiter": 12})
    rp = g.optimize(packed, options={"maxiter": 12})
    assert packed.mll > -1e300 and rp.fun == pytest.approx(rd.fun, rel=1e-6)
    np.testing.assert_allclose(rp.x, rd.x, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("world,n,block,per", [(2, 1000, None, 0), (3, 1793, None, 0), (4, 2600, None, 0), (2, 2600, 512, 0), (3, 4200, 1024, 0),
                                               (2, 3000, 256, 2)])
def test_virtual_ranks_on_one_gpu(world, n, block, per):
    x, y, xs = _problem(n)
    ln = math.log(0.1)
    shared = LocalThreadComm.Shared(world)
    errs, logs = [], {}

    def run(rank):
        try:
            ctx = g.Context(0)  # own streams per virtual rank
            comm = LocalThreadComm(shared, rank)
            comm.selftest(ctx)
            gp = gd.ShardedGPE(x, y, g.MeanConst(0.1), g.from_spec(SPEC), ln, comm=comm, ctx=ctx, block=block, stripe_blocks=per)
            assert gp.nown == len(range(rank, gp.nblk, world))
            _check(gp, x, y, xs, ln, ("const", 0.1), grad=True)
            logs[rank] = comm.log
        except BaseException as e:  # noqa: BLE001
            import traceback

            errs.append((rank, repr(e), traceback.format_exc()[-1500:]))
            shared.barrier.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert all(logs[r] == logs[0] for r in range(1, world))


@pytest.mark.parametrize("devs,n,block,per", [([0, 0], 1000, None, 0), ([0, 0, 0], 1793, None, 0), ([0, 0], 3000, 256, 2), ([0, 0, 0, 0], 5200, 512, 0)])
def test_in_process_device_group(devs, n, block, per):
    """gpmi_ctx_create(n_devices > 1): the single-process multi-GPU form.  On the one GPU of the test box the group's
    members all sit on device 0: worker threads, the in-process communicator (peer copies ordered by events) and the sharded
    driver are exercised end to end through the plain GPE verbs, gradient and full_cov included."""
    x, y, xs = _problem(n)
    ln = math.log(0.1)
    ctx = g.Context(devices=devs)
    gp = gd.ShardedGPE(x, y, g.MeanConst(0.1), g.from_spec(SPEC), ln, ctx=ctx, block=block, stripe_blocks=per)
    assert gp.nown == len(range(0, gp.nblk, len(devs)))
    _check(gp, x, y, xs, ln, ("const", 0.1), grad=True)
    hyp = gp.get_params()
    gp.set_params([h + 0.03 for h in hyp])
    gp.update_mll()
    dense = g.GP(x, y, g.MeanConst(0.1), g.from_spec(SPEC), ln)
    dense.set_params([h + 0.03 for h in hyp])
    dense.update_mll()
    assert gp.mll == pytest.approx(dense.mll, rel=1e-10)
    del gp
    ctx.close()


def test_in_process_device_group_not_posdef_and_reuse():
    n = 700
    x = np.zeros((4, n))
    x[0] = np.arange(n)
    x[:, 300] = x[:, 5]
    ctx = g.Context(devices=[0, 0])
    with pytest.raises(g.PosDefException) as ei:
        gd.ShardedGPE(x, np.ones(n), g.MeanZero(), g.SEIso(-3.0, 0.0), -400.0, ctx=ctx)
    assert ei.value.info == 301
    # the group stays usable
    x2, y2, xs2 = _problem(900)
    gp = gd.ShardedGPE(x2, y2, g.MeanZero(), g.from_spec(SPEC), math.log(0.1), ctx=ctx)
    _check(gp, x2, y2, xs2, math.log(0.1), ("zero",))


def test_virtual_ranks_not_posdef():
    n = 700
    x = np.zeros((4, n))
    x[0] = np.arange(n)
    x[:, 300] = x[:, 5]
    y = np.ones(n)
    shared = LocalThreadComm.Shared(2)
    infos, errs = [], []

    def run(rank):
        try:
            gd.ShardedGPE(x, y, g.MeanZero(), g.SEIso(-3.0, 0.0), -400.0, comm=LocalThreadComm(shared, rank), ctx=g.Context(0))
        except g.PosDefException as e:
            infos.append(e.info)
        except BaseException as e:  # noqa: BLE001
            errs.append(repr(e))
            shared.barrier.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert infos == [301, 301]


_TWO_PROC = r"""
import math, os, sys
import numpy as np
import torch, torch.distributed as dist
root = os.getcwd()
sys.path.insert(0, os.path.join(root, "gaussianprocesses.jl_amd")); sys.path.insert(0, root)
import gpmi355x as g
from gpmi355x import dist as gd
rank, world, port, n = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
torch.cuda.set_device(0)
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
rng = np.random.default_rng(11)
d = 8
x = rng.uniform(size=(d, n)); y = np.sin(2 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n); xs = rng.uniform(size=(d, 200))
ll = [math.log(0.5) + 0.05 * k for k in range(d)]
ctx = g.Context(0)
comm = gd.TorchDistComm(device=0)
comm.selftest(ctx)
gp = gd.ShardedGPE(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1), comm=comm, ctx=ctx, block=1024)
assert gp.nown == len(range(rank, gp.nblk, world)) and gp.WD == 1024
mu, s2 = gp.predict_f(xs)
_, S = gp.predict_f(xs[:, :64], full_cov=True)
gp.update_dmll()
# the single-GPU dense path on the same inputs, in this process
ref = g.GP(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1), ctx=ctx)
mu_r, s2_r = ref.predict_f(xs)
_, S_r = ref.predict_f(xs[:, :64], full_cov=True)
ref.update_dmll()
assert abs(gp.mll - ref.mll) <= 1e-10 * abs(ref.mll), (gp.mll, ref.mll)
np.testing.assert_allclose(gp.alpha, ref.alpha, rtol=1e-6, atol=1e-8 * np.abs(ref.alpha).max())
np.testing.assert_allclose(mu, mu_r, rtol=1e-7, atol=1e-9)
np.testing.assert_allclose(s2, s2_r, rtol=1e-6, atol=1e-10)
np.testing.assert_allclose(S, S_r, rtol=1e-6, atol=1e-9)
np.testing.assert_allclose(gp.dmll, ref.dmll, rtol=1e-6, atol=1e-8 * np.abs(ref.dmll).max())
# a second fit (new hyper-parameters) on the same handles
hyp = gp.get_params(); gp.set_params([h + 0.02 for h in hyp]); gp.update_mll()
ref.set_params([h + 0.02 for h in hyp]); ref.update_mll()
assert abs(gp.mll - ref.mll) <= 1e-10 * abs(ref.mll)
dist.barrier()
print(f"two-proc rank {rank} ok mll {gp.mll:.6f}", flush=True)
dist.destroy_process_group()
"""


def test_two_processes_on_one_gpu_gloo_device_buffers():
    """two PROCESSES, real torch.distributed collectives on libgpmi's device buffers (gloo moves them
    through the host; the callbacks enqueue on libgpmi's stream), N = 12 288 in 1024-row blocks, against the dense path."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [subprocess.Popen([sys.executable, "-c", _TWO_PROC, str(r), "2", str(port), "12288"], cwd=ROOT, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0 and f"two-proc rank {r} ok" in o, (o[-1500:], e[-3000:])


def test_bench_multi_gpu_line_rehearsed_on_one_gpu(tmp_path):
    """The driver's multi-GPU tier runs `bench.py --gpus N` once, at the end of the round, on hardware no round has had: the whole world > 1
    branch of bench.py (torchrun launch, sharded fit, max-over-ranks timing, the ONE line with `parity`, `per_step_ms` and `c4_sharded`,
    the primary line persisted before the secondary objects, every rank leaving with exit code 0) is rehearsed here on one GPU
    (--dry-run-one-gpu: two processes on the two CU partitions, gloo through the host).  Round 6 rewrote that branch and the first
    rehearsal crashed rank 1 (`out` exists on rank 0 only) after the line was out — torchrun then reports failure for the whole job."""
    import json

    env = dict(os.environ)
    env["GPMI_BENCH_SECONDARY_S"] = "600"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-one-gpu", "--n", "8192", "--c4-n", "12288", "--steps", "1",
                          "--warmup", "1", "--secondary", "c4"], cwd=ROOT, capture_output=True, text=True, timeout=1200, env=env)
    lines = [ln_ for ln_ in out.stdout.splitlines() if ln_.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.returncode, out.stdout[-1500:], out.stderr[-3000:])
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["scaling"] == "strong"
    assert j["parity"]["checked"] and j["parity"]["ok"], j["parity"]
    assert j["per_step_ms"] and j["c4_sharded"]["parity"]["ok"], j.get("c4_sharded")
    for f in ("bench_gpus2_primary.json", "bench_gpus2.json"):
        assert os.path.exists(os.path.join(ROOT, "gpurun_out", f)), f


def test_rccl_native_communicator_in_a_group_of_one():
    """libgpmi's own RCCL communicator (librccl opened at run time, ncclCommInitRank / Broadcast / AllGather / AllReduce) with one
    rank: gpmi_comm_selftest, then a blocked fit with it (the collectives are skipped at world 1, the handle plumbing is not)."""
    ctx = g.Context(0)
    comm = gd.RcclComm(ctx, 0, 1, exchange_id=lambda b: b)
    comm.selftest(ctx)
    x, y, xs = _problem(1300)
    ln = math.log(0.1)
    gp = gd.ShardedGPE(x, y, g.MeanZero(), g.from_spec(SPEC), ln, comm=comm, ctx=ctx)
    _check(gp, x, y, xs, ln, ("zero",))


def test_torch_nccl_backend_behind_the_callbacks_in_a_group_of_one():
    """torch.distributed's "nccl" backend (= RCCL) through TorchDistComm's callbacks: tensor views of raw device pointers, the
    external stream, every collective — gpmi_comm_selftest in a one-rank group (its own process group, so a subprocess)."""
    code = r"""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.getcwd(), "gaussianprocesses.jl_amd")); sys.path.insert(0, os.getcwd())
import gpmi355x as g
from gpmi355x import dist as gd
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ctx = g.Context(0)
comm = gd.TorchDistComm(device=0)
comm.selftest(ctx)
rc = gd.rccl_comm(ctx)       # the unique id through broadcast_object_list, then libgpmi's own RCCL communicator
rc.selftest(ctx)
print("nccl-callbacks ok", comm.error)
dist.destroy_process_group()
"""
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "nccl-callbacks ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


_OVERLAP = r"""
import json, math, os, sys, threading, time
import numpy as np
root = os.getcwd()
sys.path.insert(0, os.path.join(root, "gaussianprocesses.jl_amd")); sys.path.insert(0, root)
import gpmi355x as g
from gpmi355x import dist as gd
n, WD, d = 65536, 1024, 8
rng = np.random.default_rng(17)
x = rng.uniform(size=(d, n)); y = np.sin(2 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
ll = [math.log(0.5) + 0.05 * k for k in range(d)]; ln = math.log(0.1)
kern = lambda: g.SEArd(ll, 0.0)
# ---- (1) two CU partitions work CONCURRENTLY: two fits side by side take about as long as one of them
halves = [g.GP(x[:, :16384], y[:16384], g.MeanZero(), kern(), ln, ctx=g.Context(256 * (1 + p))) for p in range(2)]
def timed(models):
    ts = [threading.Thread(target=m.update_mll) for m in models]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    return (time.perf_counter() - t0) * 1e3
timed(halves)
one = min(timed(halves[:1]) for _ in range(3)); both = min(timed(halves) for _ in range(3))
assert abs(halves[0].mll - halves[1].mll) <= 1e-12 * abs(halves[0].mll)
del halves
# ---- (2) a model sharded over the two partitions, with injected latency in front of the collectives
ctx = g.Context(devices=[256, 512])
gp = gd.ShardedGPE(x, y, g.MeanZero(), kern(), ln, ctx=ctx, block=WD)
mll0 = gp.mll
def fit_ms(m):
    ts = []
    for rep in range(2):
        t0 = time.perf_counter(); m.update_mll(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
res = {}
t0 = fit_ms(gp)
# the panel exchange hides under U2b = everything of a step's update that U2a (sized to cover the chain kernel + the inverse broadcast,
# blocked.cpp u2a_cover_s_) does not take
os.environ["GPMI_TEST_COMM_DELAY_ON"] = "2"; os.environ["GPMI_TEST_COMM_DELAY_US"] = "5000"
res["panel_exchange+5ms"] = fit_ms(gp)
assert abs(gp.mll - mll0) <= 1e-12 * abs(mll0)
os.environ.pop("GPMI_TEST_COMM_DELAY_US"); os.environ.pop("GPMI_TEST_COMM_DELAY_ON")
t0_again = fit_ms(gp)
del gp
# a SLOW inverse broadcast is hidden by telling the driver how long it is (GPMI_BLOCKED_U2A_US = chain + broadcast: U2a grows to cover it):
# what a launcher does with the `broadcast` phase time of bench.py's per_step_ms
os.environ["GPMI_BLOCKED_U2A_US"] = "6500"
gp = gd.ShardedGPE(x, y, g.MeanZero(), kern(), ln, ctx=ctx, block=WD)
t0_wide = fit_ms(gp)
os.environ["GPMI_TEST_COMM_DELAY_ON"] = "1"; os.environ["GPMI_TEST_COMM_DELAY_US"] = "5000"
res["inverse_broadcast+5ms"] = fit_ms(gp)
assert abs(gp.mll - mll0) <= 1e-12 * abs(mll0)
os.environ.pop("GPMI_TEST_COMM_DELAY_US"); os.environ.pop("GPMI_TEST_COMM_DELAY_ON"); os.environ.pop("GPMI_BLOCKED_U2A_US")
print("OVERLAP " + json.dumps({"n": n, "block": WD, "one_fit_on_a_partition_ms": one, "two_fits_side_by_side_ms": both,
                               "fit_ms_no_delay": t0, "fit_ms_no_delay_again": t0_again, "fit_ms_no_delay_u2a_6500us": t0_wide, "fit_ms": res}), flush=True)
"""


def test_cu_partitions_and_injected_collective_latency():
    """VERDICT r3 Next 1(a, b).  (a) The driver refused compute partitioning of the leased MI355X (profiles/r04_a_cpx_refused.log), so
    the two logical devices are the two CU PARTITIONS of the one GPU (device ids 256 and 512: disjoint halves of every XCD; include/
    gpmi.h) — the test first shows that they really run side by side (two fits at once take well under twice one).  (b) Does the
    one-step look-ahead hide the exchange?  A model of N = 65 536 sharded over the two partitions (in-process device group, libgpmi's
    event-ordered peer-copy communicator); a test hook (GPMI_TEST_COMM_DELAY_US / _ON) puts 5 ms of extra latency — a spin kernel on
    the stream the collective is given — in front of every panel exchange (63 x 5 = 315 ms injected), or of every inverse broadcast.
    A serial exchange would lengthen the fit by the whole injected total.  The pipeline of csrc/blocked.cpp hides the exchange under
    U2b — everything of step k's update (t_k ~ (rows left)^2) that U2a does not take — and the broadcast under U2a, which round 5 sizes
    in flops to cover just the ~1.1 ms chain kernel + a broadcast (u2a_cover_s_; a launcher that measures a slow broadcast widens it:
    GPMI_BLOCKED_U2A_US, used below for the broadcast case).  Both must come out well below 1.0 — and cannot reach 0 (the last steps
    are shorter than the delay).  Timing-sensitive, so it runs in a process of its own (a long-lived pytest
    process that has created dozens of contexts shares hardware queues between their streams) and gets TWO attempts: the bounds are wall-clock
    ratios, and one noisy box must not take the rest of a `pytest -x` run down with it (a regression of the pipeline fails both)."""
    try:
        _overlap_attempt()
    except AssertionError as first:
        print("overlap test: first attempt failed, retrying once:", str(first)[:2000])
        _overlap_attempt()


def _overlap_attempt():
    import json

    out = subprocess.run([sys.executable, "-c", _OVERLAP], cwd=ROOT, capture_output=True, text=True, timeout=1500)
    line = [ln_ for ln_ in out.stdout.splitlines() if ln_.startswith("OVERLAP ")]
    assert out.returncode == 0 and line, out.stdout[-1500:] + out.stderr[-3000:]
    r = json.loads(line[0][8:])
    assert r["two_fits_side_by_side_ms"] < 1.7 * r["one_fit_on_a_partition_ms"], r   # (serial would be 2.0; measured 1.43 - 1.46: shared L2 / HBM / clocks)
    n, WD, t0 = r["n"], r["block"], min(r["fit_ms_no_delay"], r["fit_ms_no_delay_again"])
    nblk = -(-n // WD)
    rem2 = np.array([(n - (k + 1) * WD) ** 2 for k in range(nblk - 1)], dtype=float)
    tk = 0.5 * t0 * rem2 / rem2.sum()                # step k's update, taking the updates as half of the fit (measured: ~0.5 on partitions)
    r["delays"] = {}
    for key, t in r["fit_ms"].items():
        name, D = key.split("+")[0], float(key.split("+")[1][:-2])
        if name == "panel_exchange":   # hidden under U2b = the step's update minus U2a's ~1.5 ms
            count, base = nblk - 1, t0
            room = np.maximum(tk - 1.5, 0.0)
        else:                          # hidden under U2a, widened to 6.5 ms for the purpose (minus the ~1.1 ms chain in front of it)
            count, base = nblk, r["fit_ms_no_delay_u2a_6500us"]
            room = np.concatenate(([0.0], np.minimum(tk, 6.5) - 1.1))
        injected = D * count
        uncover = float(np.sum(np.maximum(D - room[:count], 0.0)))
        r["delays"][key] = {"fit_ms": t, "baseline_ms": base, "injected_ms": injected, "extra_ms": t - base, "exposed_fraction": (t - base) / injected,
                            "model_uncoverable_fraction": uncover / injected}
    print("injected-latency overlap:", json.dumps(r))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r05_overlap_latency.json"), "w") as fh:
        json.dump(r, fh, indent=1)
    # A serial exchange / broadcast: 1.0.  Round 4: 0.60 / 0.33 before the owner took its next diagonal block from its own rows, 0.26 / 0.28
    # after — with the 3.5 ms multi-launch chain exposed IN PARALLEL with the injected delay in the late steps, which flattered the ratio:
    # the same tree with the 1.1 ms chain kernel measures 0.70 / 0.29 - 0.38 on a 4.5 % faster fit (profiles/r05_g_*: both end at 2105 ms with
    # the delay).  What the pipeline itself hides is bounded by the step model above (room under U2b / U2a per step): the measured exposure
    # must stay within 0.15 of the model's uncoverable fraction — a regression of the pipeline (exchange serialised behind the update, chain
    # waiting for the gather) adds far more than that.
    for key in ("panel_exchange+5ms", "inverse_broadcast+5ms"):
        d = r["delays"][key]
        assert d["exposed_fraction"] < d["model_uncoverable_fraction"] + 0.15, r
        assert d["exposed_fraction"] < 0.75, r


def test_cu_partitions_sharded_model_matches_the_oracle():
    """a device group over the two CU partitions of device 0 through the plain GPE verbs (gradient, full_cov, PDMat surface)"""
    x, y, xs = _problem(6000)
    ln = math.log(0.1)
    ctx = g.Context(devices=[256, 512])
    gp = gd.ShardedGPE(x, y, g.MeanConst(0.1), g.from_spec(SPEC), ln, ctx=ctx)
    _check(gp, x, y, xs, ln, ("const", 0.1), grad=True)
    del gp
    ctx.close()
