"""Timing of update_mll! (+ predict_f) under environment knobs that are read once per context (tools only): one fresh process per
configuration would pay the first-import minutes every time, so the knobs are set in os.environ and a NEW context is created for each.
Usage: python tools/knob_sweep.py N [blocked] -- prints fit / predict ms (min of 3) and the mll for every configuration."""
import math, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussianprocesses.jl_amd"))
import numpy as np
import gpmi355x as g
from gpmi355x import dist as gd

KNOBS = ("GPMI_SUPER", "GPMI_CUMASK_BELOW", "GPMI_LOOKAHEAD_MIN", "GPMI_TAIL_FUSE", "GPMI_CHAIN", "GPMI_CHAIN_WGS", "GPMI_UPDATE256_MIN", "GPMI_LOOKAHEAD", "GPMI_CHAIN_BESIDE_WGS", "GPMI_SUPER_W")

def synth(n, d, p, seed=20240501):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0.0, 1.0, size=(d, n)); y = np.sin(2*np.pi*x).sum(axis=0)/d + 0.1*rng.standard_normal(n)
    return x, y, rng.uniform(0.0, 1.0, size=(d, p))

def run(n, env, blocked=None, reps=3, d=8, p=1024):
    for k in KNOBS: os.environ.pop(k, None)
    for k, v in env.items(): os.environ[k] = str(v)
    ctx = g.Context(0)
    x, y, xs = synth(n, d, p)
    ll = [math.log(0.5) + 0.05*k for k in range(d)]
    if blocked is not None:
        gp = gd.ShardedGPE(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1), ctx=ctx, block=blocked or None)
    else:
        gp = g.GP(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1), ctx=ctx)
    ts = []
    for i in range(reps):
        hyp = np.asarray(gp.get_params()); hyp[1:] += 1e-3
        a = time.perf_counter(); gp.set_params(hyp); gp.update_mll(); b = time.perf_counter(); mu, s2 = gp.predict_f(xs); c = time.perf_counter()
        ts.append((b - a, c - b))
    fit = min(t[0] for t in ts); pred = min(t[1] for t in ts)
    tag = " ".join(f"{k[5:]}={v}" for k, v in env.items()) or "defaults"
    print(f"N={n}{' blocked=' + str(blocked) if blocked is not None else ''} [{tag}]: fit {fit*1e3:.2f} ms  predict {pred*1e3:.2f} ms  step {1e3*(fit+pred):.2f}  mll={gp.mll:.9g}", flush=True)
    del gp
    ctx.close()

if __name__ == "__main__":
    n = int(sys.argv[1])
    mode = sys.argv[2] if len(sys.argv) > 2 else "dense"
    if mode == "dense":
        cfgs = [{}, {"GPMI_TAIL_FUSE": 0}, {"GPMI_CHAIN": 0},
                {"GPMI_SUPER": "4096,12288,24576"}, {"GPMI_SUPER": "2048,8192,24576"}, {"GPMI_SUPER": "2048,6144,24576"}, {"GPMI_SUPER": "2048,4096,24576"},
                {"GPMI_SUPER": "2048,4096,16384"}, {"GPMI_SUPER": "1024,4096,24576"}, {"GPMI_SUPER": "2048,6144,16384"},
                {"GPMI_CUMASK_BELOW": 0}, {"GPMI_CUMASK_BELOW": 16384}, {"GPMI_CUMASK_BELOW": 0, "GPMI_SUPER": "2048,6144,24576"},
                {"GPMI_LOOKAHEAD_MIN": 2048}, {"GPMI_LOOKAHEAD_MIN": 1536, "GPMI_SUPER": "2048,6144,24576"},
                {"GPMI_TAIL_FUSE": 1024}, {"GPMI_TAIL_FUSE": 1536}, {"GPMI_UPDATE256_MIN": 512}, {"GPMI_UPDATE256_MIN": 256, "GPMI_CUMASK_BELOW": 0}]
        for e in cfgs: run(n, e)
    elif mode == "r6":  # round 6: the thresholds again, with potf2_wg and the update's full grid (profiles/r06_i_*)
        for e in ({}, {"GPMI_SUPER": "2048,6144,16384"}, {"GPMI_SUPER": "2048,6144,11264"}, {"GPMI_SUPER": "2048,6144,9216"}, {"GPMI_SUPER": "2048,4096,13312"},
                  {"GPMI_SUPER": "2048,8192,13312"}, {"GPMI_SUPER": "1024,6144,13312"}, {"GPMI_SUPER": "3072,6144,13312"}, {"GPMI_SUPER": "2048,4096,9216"},
                  {"GPMI_CUMASK_BELOW": 0}, {"GPMI_CUMASK_BELOW": 16384}, {"GPMI_CUMASK_BELOW": 24576}, {"GPMI_TAIL_FUSE": 1024}, {"GPMI_TAIL_FUSE": 1536}, {}):
            run(n, e)
    elif mode == "la":  # round 6, after the chain got sixteen workgroups beside the update: where the look-ahead stops paying (profiles/r06_n_*)
        for e in ({}, {"GPMI_LOOKAHEAD_MIN": 2560}, {"GPMI_LOOKAHEAD_MIN": 3072}, {"GPMI_LOOKAHEAD_MIN": 3584}, {"GPMI_LOOKAHEAD_MIN": 4096}, {"GPMI_LOOKAHEAD_MIN": 5632},
                  {"GPMI_LOOKAHEAD_MIN": 3072, "GPMI_SUPER": "2048,4096,13312"}, {"GPMI_SUPER": "2048,4096,13312"}, {}):
            run(n, e)
    elif mode == "mask":  # round 6: below which size whole (CU-masked) compute units for the chain still beat the full-grid update with the chain placed first
        for e in ({}, {"GPMI_CUMASK_BELOW": 0}, {"GPMI_CUMASK_BELOW": 8192}, {"GPMI_CUMASK_BELOW": 12288}, {}, {"GPMI_CUMASK_BELOW": 0}):
            run(n, e)
    elif mode == "misc":  # round 6: free slots beside the 128 x 128 updates, the 256 x 128 kernel's minimum launch, chain workgroups beside it
        for e in ({}, {"GPMI_LOOKAHEAD": 8}, {"GPMI_LOOKAHEAD": 24}, {"GPMI_LOOKAHEAD": 32}, {"GPMI_UPDATE256_MIN": 768}, {"GPMI_UPDATE256_MIN": 512}, {"GPMI_UPDATE256_MIN": 1536},
                  {"GPMI_CHAIN_BESIDE_WGS": 24}, {}):
            run(n, e)
    elif mode == "wide":  # round 6: the widest super-panel class at 1536 / 1792 columns instead of 2048 (panel solves ~ W, the update's C passes ~ 1 / W)
        for e in ({}, {"GPMI_SUPER_W": 1536}, {"GPMI_SUPER_W": 1792}, {"GPMI_SUPER_W": 1280}, {}, {"GPMI_SUPER_W": 1536}):
            run(n, e)
    elif mode == "fine":
        for sup in ("2048,6144,16384", "2048,6144,12288", "2048,4096,12288", "1024,4096,12288", "2048,5120,10240", "1536,4096,8192"):
            for below in (32768, 0):
                run(n, {"GPMI_SUPER": sup, "GPMI_CUMASK_BELOW": below})
    else:
        for blk in (0, 1024, 2048):
            for e in ({}, {"GPMI_CHAIN": 0}):
                if blk == 0 and e: continue
                run(n, e, blocked=blk)
