"""Functional check of the multi-device paths on DISTINCT logical devices (VERDICT r3 "Next 1a"): run only when the box shows more than one
device (an 8-GPU node, or one MI355X in CPX compute-partition mode = 8 logical devices of one XCD each).

  python tools/multidev_check.py group <ndev> [n]      in-process device group Context(devices=[0..ndev-1]) vs the dense path
  python -m torch.distributed.run --nproc-per-node W tools/multidev_check.py ranks [n]
                                                       one process per device: libgpmi's RCCL communicator + torch nccl callbacks,
                                                       selftests, then a sharded fit / predict / gradient vs the dense path
Parity, not speed: exit code 0 means every number agreed.
"""
import math
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "gaussianprocesses.jl_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def problem(n, d=8, p=200, seed=11):
    rng = np.random.default_rng(seed)
    x = rng.uniform(size=(d, n))
    y = np.sin(2 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
    xs = rng.uniform(size=(d, p))
    ll = [math.log(0.5) + 0.05 * k for k in range(d)]
    return x, y, xs, ll


def compare(g, gp, ref, xs, tag):
    mu, s2 = gp.predict_f(xs)
    _, S = gp.predict_f(xs[:, :64], full_cov=True)
    gp.update_dmll()
    mu_r, s2_r = ref.predict_f(xs)
    _, S_r = ref.predict_f(xs[:, :64], full_cov=True)
    ref.update_dmll()
    assert abs(gp.mll - ref.mll) <= 1e-10 * abs(ref.mll), (tag, gp.mll, ref.mll)
    np.testing.assert_allclose(gp.alpha, ref.alpha, rtol=1e-6, atol=1e-8 * np.abs(ref.alpha).max())
    np.testing.assert_allclose(mu, mu_r, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(s2, s2_r, rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(S, S_r, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(gp.dmll, ref.dmll, rtol=1e-6, atol=1e-8 * np.abs(ref.dmll).max())
    return dict(mll=gp.mll, mll_dense=ref.mll, dmu=float(np.abs(mu - mu_r).max()), ds2=float(np.abs(s2 - s2_r).max()),
                dgrad=float(np.abs(gp.dmll - ref.dmll).max()))


def main():
    mode = sys.argv[1]
    if mode == "group":
        import gpmi355x as g
        from gpmi355x import dist as gd

        ndev = int(sys.argv[2])
        n = int(sys.argv[3]) if len(sys.argv) > 3 else 12288
        x, y, xs, ll = problem(n)
        ctx = g.Context(devices=list(range(ndev)))
        gp = gd.ShardedGPE(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1), ctx=ctx, block=1024 if n >= 8192 else None)
        ref = g.GP(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1), ctx=g.Context(0))
        r = compare(g, gp, ref, xs, "group")
        print(f"multidev group ok: {ndev} DISTINCT devices, n={n}, blocks of {gp.WD}: {r}", flush=True)
        return
    # one process per device
    import torch
    import torch.distributed as dist

    rank, world, lr = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 12288
    torch.cuda.set_device(lr)
    dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
    import gpmi355x as g
    from gpmi355x import dist as gd

    ctx = g.Context(lr)
    x, y, xs, ll = problem(n)
    out = {}
    for kind in ("rccl", "torch"):
        comm = gd.rccl_comm(ctx) if kind == "rccl" else gd.TorchDistComm(device=lr)
        comm.selftest(ctx)
        gp = gd.ShardedGPE(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1), comm=comm, ctx=ctx, block=1024 if n >= 8192 else None)
        ref = g.GP(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1), ctx=ctx)
        out[kind] = compare(g, gp, ref, xs, kind)
        del gp, ref
        dist.barrier()
    if rank == 0:
        print(f"multidev ranks ok: world={world} on distinct devices, n={n}: {out}", flush=True)
    dist.destroy_process_group()
    sys.stdout.flush()
    os._exit(0)


if __name__ == "__main__":
    main()
