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


@pytest.mark.parametrize("n", [300, 1000, 1793])
def test_single_rank_device_ops(n):
    x, y, xs = _problem(n)
    ln = math.log(0.1)
    gp = gd.ShardedGPE(x, y, g.MeanConst(0.1), g.from_spec(SPEC), ln)
    _check(gp, x, y, xs, ln, ("const", 0.1))


@pytest.mark.parametrize("world,n", [(2, 1000), (3, 1793), (4, 2600)])
def test_virtual_ranks_on_one_gpu(world, n):
    x, y, xs = _problem(n)
    ln = math.log(0.1)
    shared = LocalThreadComm.Shared(world)
    errs = []

    def run(rank):
        try:
            ctx = g.Context(0)  # own stream per virtual rank
            gp = gd.ShardedGPE(x, y, g.MeanConst(0.1), g.from_spec(SPEC), ln, comm=LocalThreadComm(shared, rank), ctx=ctx)
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
