"""N > 1 path on CPU: the sharded orchestration (gpmi355x.dist) under world_size-2 gloo, with the device ops
replaced by a NumPy stand-in (tests/dist_helpers.FakeOps), checked against the oracle.  What this pins:
block-cyclic ownership, the broadcast of the diagonal block's inverse, the panel all-gather + scatter into global row
order, the look-ahead split of the update on the owner of the next block, the carried right-hand-side row, the
distributed backward solve (partial sums + per-block all-reduce), logdet all-reduce, the split of test points in
predict_f and the PosDefException contract across ranks."""
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, case):
    for p in (ROOT, os.path.join(ROOT, "gaussianprocesses.jl_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import gpmi355x as g
    from gpmi355x import dist as gd
    from dist_helpers import FakeOps
    from oracle import gp_oracle as G

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)
        d = 3
        x = rng.uniform(size=(d, n))
        y = np.sin(3 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
        xs = rng.uniform(size=(d, 37))
        spec = ("sum", ("se_ard", [-0.5, -0.3, -0.6], 0.2), ("mat52_iso", -0.4, -0.5))
        ln = math.log(0.1)
        comm = gd.TorchDistComm()
        if case == "notpd":
            # far-apart points (K ~ I) with one exact duplicate inside block 1 (owned by rank 1): pivot 301 is
            # exactly 1 - 1 = 0 for dpotrf and for the sharded factorisation alike
            x = np.zeros((d, n))
            x[0] = np.arange(n)
            x[:, 300] = x[:, 5]
            nspec = ("se_iso", -3.0, 0.0)
            with pytest.raises(g.PosDefException) as ei:
                gd.ShardedGPE(x, y, g.MeanZero(), g.from_spec(nspec), -400.0, comm=comm, ops=FakeOps(nspec))
            with pytest.raises(G.NotPosDef) as eo:
                G.update_mll(nspec, x, y, -400.0)
            assert ei.value.info == eo.value.info == 301
            return
        block = 512 if case == "fit512" else None  # the super-panel as the distributed block: 512 rows = 4 tiles per block
        stripes = 2 if case == "packed" else None   # packed storage: stripes of 2 local blocks, no upper triangle
        gp = gd.ShardedGPE(x, y, g.MeanConst(0.2), g.from_spec(spec), ln, comm=comm, ops=FakeOps(spec), block=block,
                           stripe_blocks=stripes)
        assert gp.WD == (block or 256)
        if stripes:
            assert len(gp.S.items) > 1 and gp.S.items[0][2].shape[1] < gp.npad
        ref = G.update_mll(spec, x, y, ln, ("const", 0.2))
        assert abs(gp.mll - ref["mll"]) <= 1e-9 * abs(ref["mll"]), (gp.mll, ref["mll"])
        np.testing.assert_allclose(gp.alpha, ref["alpha"], rtol=1e-7, atol=1e-9)
        assert abs(gp.logdet - ref["logdet"]) <= 1e-9 * abs(ref["logdet"])
        mu, s2 = gp.predict_f(xs)
        mu_o, s2_o = G.predict_f(spec, x, ref, xs, ("const", 0.2))
        np.testing.assert_allclose(mu, mu_o, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(s2, s2_o, rtol=1e-6, atol=1e-10)
        # ownership really is split: this rank holds only its share of the factor
        assert gp.nown == len(range(rank, gp.nblk, world)) and (stripes or gp.A.shape[0] == gp.nown * gp.WD + 8)
        # refit with new hyper-parameters reuses the buffers
        hyp = gp.get_params()
        gp.set_params([h + 0.05 for h in hyp])
        spec2 = ("sum", ("se_ard", [-0.45, -0.25, -0.55], 0.25), ("mat52_iso", -0.35, -0.45))
        gp.ops.spec = spec2  # the stand-in ops evaluate the oracle spec, keep it in step with the kernel object
        assert gp.kernel.get_params() == pytest.approx([-0.45, -0.25, -0.55, 0.25, -0.35, -0.45])
        gp.update_target()
        ref2 = G.update_mll(spec2, x, y, ln + 0.05, ("const", 0.25))
        assert abs(gp.mll - ref2["mll"]) <= 1e-9 * abs(ref2["mll"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 700), (2, 1100), (3, 1300)])
def test_sharded_fit_predict_gloo(world, n):
    mp.spawn(_worker, args=(world, _free_port(), n, "fit"), nprocs=world, join=True)


def test_sharded_fit_predict_gloo_512_row_blocks():
    mp.spawn(_worker, args=(2, _free_port(), 1900, "fit512"), nprocs=2, join=True)


@pytest.mark.parametrize("world,n", [(2, 2100), (3, 2500)])
def test_sharded_packed_stripes_gloo(world, n):
    """the owned block-rows in stripes that stop at their own diagonal (no upper triangle allocated)."""
    mp.spawn(_worker, args=(world, _free_port(), n, "packed"), nprocs=world, join=True)


def test_sharded_not_posdef_gloo():
    mp.spawn(_worker, args=(2, _free_port(), 600, "notpd"), nprocs=2, join=True)
