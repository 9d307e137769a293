"""Sharded path on the GPU: the real HIP building blocks (gpmi_dev_*) under the row-block-cyclic
orchestration, (a) as a single rank and (b) as G virtual ranks (threads, each with its own gpmi context and
stream) sharing the one GPU of the test box through an in-process communicator.  Together with
tests/test_dist_cpu.py (real torch.distributed collectives, stand-in ops) this covers both halves of the
multi-GPU path; the RCCL transport itself is torch.distributed's."""
import math
import threading

import numpy as np
import pytest

import gpmi355x as g
from gpmi355x import dist as gd
from dist_helpers import LocalThreadComm
from oracle import gp_oracle as G

pytestmark = pytest.mark.gpu

SPEC = ("sum", ("se_ard", [-0.5, -0.3, -0.6, -0.2], 0.2), ("mat52_iso", -0.4, -0.5))


def _problem(n, p=45, seed=3):
    rng = np.random.default_rng(seed)
    x = rng.uniform(size=(4, n))
    y = np.sin(3 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
    xs = rng.uniform(size=(4, p))
    return x, y, xs


def _check(gp, x, y, xs, ln, mspec, rtol_mll=1e-10):
    ref = G.update_mll(SPEC, x, y, ln, mspec)
    assert abs(gp.mll - ref["mll"]) <= rtol_mll * abs(ref["mll"]), (gp.mll, ref["mll"])
    np.testing.assert_allclose(gp.alpha, ref["alpha"], rtol=1e-7, atol=1e-8 * np.abs(ref["alpha"]).max())
    assert abs(gp.logdet - ref["logdet"]) <= 1e-10 * abs(ref["logdet"])
    mu, s2 = gp.predict_f(xs)
    mu_o, s2_o = G.predict_f(SPEC, x, ref, xs, mspec)
    np.testing.assert_allclose(mu, mu_o, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(s2, s2_o, rtol=1e-6, atol=1e-10)


@pytest.mark.parametrize("n,block", [(300, None), (1000, None), (1793, None), (5000, None), (3000, 1024), (4500, 2048)])
def test_single_rank_device_ops(n, block):
    """block None: 256-row blocks below 4096 points, 512 from there (dist.default_block); the look-ahead — next diagonal
    block updated first, factored and inverted on the side stream under the rest of the update — runs at every step."""
    x, y, xs = _problem(n)
    ln = math.log(0.1)
    gp = gd.ShardedGPE(x, y, g.MeanConst(0.1), g.from_spec(SPEC), ln, block=block)
    assert gp.WD == (block or (512 if n >= 4096 else 256))
    _check(gp, x, y, xs, ln, ("const", 0.1))


@pytest.mark.parametrize("n,block,per", [(2300, 256, 2), (5000, 512, 3), (9000, 1024, 2)])
def test_single_rank_packed_stripes(n, block, per):
    """SURVEY §8f-3 on the device: stripes that stop at their own diagonal; every update is one launch per stripe."""
    x, y, xs = _problem(n)
    ln = math.log(0.1)
    gp = gd.ShardedGPE(x, y, g.MeanConst(0.1), g.from_spec(SPEC), ln, block=block, stripe_blocks=per)
    assert len(gp.S.items) > 1 and gp.S.items[0][2].shape[1] < gp.npad
    _check(gp, x, y, xs, ln, ("const", 0.1))


@pytest.mark.parametrize("world,n,block", [(2, 1000, None), (3, 1793, None), (4, 2600, None), (2, 2600, 512), (3, 4200, 1024)])
def test_virtual_ranks_on_one_gpu(world, n, block):
    x, y, xs = _problem(n)
    ln = math.log(0.1)
    shared = LocalThreadComm.Shared(world)
    errs = []

    def run(rank):
        try:
            ctx = g.Context(0)  # own stream per virtual rank
            gp = gd.ShardedGPE(x, y, g.MeanConst(0.1), g.from_spec(SPEC), ln, comm=LocalThreadComm(shared, rank), ctx=ctx, block=block)
            assert gp.nown == len(range(rank, gp.nblk, world))
            _check(gp, x, y, xs, ln, ("const", 0.1))
        except BaseException as e:  # noqa: BLE001
            errs.append((rank, repr(e)))
            shared.barrier.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


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


def test_rccl_collectives_in_a_group_of_one():
    """The RCCL calls themselves (broadcast / all_gather_into_tensor / all_reduce on the library's device buffers) with the
    nccl backend in a single-rank group, collectives forced: everything about the real multi-GPU transport that one GPU can
    exercise.  Runs in a subprocess (its own process group)."""
    import os
    import subprocess
    import sys

    code = r"""
import math, os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.getcwd(), "gaussianprocesses.jl_amd")); sys.path.insert(0, os.getcwd())
import gpmi355x as g
from gpmi355x import dist as gd
from oracle import gp_oracle as G
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
rng = np.random.default_rng(3)
n = 1300
x = rng.uniform(size=(4, n)); y = np.sin(3 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n); xs = rng.uniform(size=(4, 33))
spec = ("sum", ("se_ard", [-0.5, -0.3, -0.6, -0.2], 0.2), ("mat52_iso", -0.4, -0.5))
gp = gd.ShardedGPE(x, y, g.MeanZero(), g.from_spec(spec), math.log(0.1), comm=gd.TorchDistComm(force=True))
ref = G.update_mll(spec, x, y, math.log(0.1))
assert abs(gp.mll - ref["mll"]) <= 1e-10 * abs(ref["mll"]), (gp.mll, ref["mll"])
mu, s2 = gp.predict_f(xs)
mu_o, s2_o = G.predict_f(spec, x, ref, xs)
np.testing.assert_allclose(mu, mu_o, rtol=1e-7, atol=1e-9); np.testing.assert_allclose(s2, s2_o, rtol=1e-6, atol=1e-10)
print("rccl-one-rank ok", gp.mll)
dist.destroy_process_group()
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "rccl-one-rank ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
