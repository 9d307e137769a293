"""Task timeline of the persistent chain kernel (csrc/chain.hip) from clock marks inside the kernel — a measurement of WHERE one launch's
time goes, which event timing around the launch cannot give.  Needs tools/bin/libgpmi_chain_trace.so (tools/build_chain_trace.sh: the
product's objects with chain.hip recompiled under -DGPMI_CHAIN_TRACE); the product library carries no marks.

Usage: python tools/chain_trace.py [--skip s] [--launches l] N dense|blocked [[--skip ..] N mode ...]
  per configuration one update_mll! warm-up, then the traced one: chain launches s .. s + l - 1 of it write 8 words per task
  (100 MHz clock: task start, first slab in LDS, K loop done, potf2 done / inverse tile in LDS, published; wait ticks; workgroup; tile).
Output: per launch the span, the busy fraction of the workgroups, the per-class phase means and the diagonal (critical-path) hand-offs."""
import math, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussianprocesses.jl_amd"))
import numpy as np

TICK_US = 0.01  # s_memrealtime: 100 MHz


def parse(path):
    launches, cur = [], None
    for line in open(path):
        w = line.split()
        if w[0] == "launch":
            cur = {"nb": int(w[2]), "wgs": int(w[4]), "ld": int(w[6]), "inverse": int(w[8]), "beside": int(w[10]), "elem": int(w[12]), "potf2": [int(q) for q in w[14:18]], "tasks": []}
            launches.append(cur)
        else:
            v = [int(q) for q in w]
            cur["tasks"].append({"t": v[0], "m": v[1:6], "wait": v[6], "wg": v[7] >> 8, "cu": v[7] & 0xff, "is_x": (v[8] >> 16) & 1, "i": (v[8] >> 8) & 0xff, "j": v[8] & 0xff})
    return launches


BRIEF = False
KEEP = None   # directory that keeps the raw mark files


def report(L, out=sys.stdout):
    tasks = [k for k in L["tasks"] if k["m"][4] != 0]
    if not tasks:
        print("  (no task recorded)", file=out); return
    t0 = min(k["m"][0] for k in tasks); t1 = max(k["m"][4] for k in tasks)
    span = (t1 - t0) * TICK_US
    busy = sum(k["m"][4] - k["m"][0] for k in tasks) * TICK_US
    wait = sum(k["wait"] for k in tasks) * TICK_US
    print(f"launch: nb {L['nb']} workgroups {L['wgs']} ld {L['ld']} inverse {L['inverse']} beside_update {L['beside']} elem {L['elem']}: "
          f"span {span:.1f} us, tasks {len(tasks)}, task time {busy:.0f} us = {busy / span / L['wgs']:.2f} of workgroups x span, in flag waits {wait:.0f} us "
          f"({wait / busy:.2f} of task time)", file=out)
    if BRIEF: return
    if L.get("potf2"):
        q = [v * TICK_US for v in L["potf2"]]
        print(f"  the last potf2 of the launch (potf2_wg, round 6: the marks cost ~1 us each): four panels and the inverse work under them {q[0]:.2f} us | last step "
              f"(X_3b, pivot test) {q[3]:.2f}", file=out)
    wgs = sorted({k["wg"] for k in tasks})
    print(f"  workgroups that took tasks: {len(wgs)}; tasks per workgroup min/max {min(sum(1 for k in tasks if k['wg'] == w) for w in wgs)}/"
          f"{max(sum(1 for k in tasks if k['wg'] == w) for w in wgs)}", file=out)
    # per class: start -> first slab in LDS -> K loop done -> (potf2 | inverse tile) -> published
    for name, sel in (("L diagonal", lambda k: not k["is_x"] and k["i"] == k["j"]), ("L below", lambda k: not k["is_x"] and k["i"] != k["j"]), ("X", lambda k: k["is_x"])):
        ks = [k for k in tasks if sel(k)]
        if not ks: continue
        def mean(f): return sum(f(k) for k in ks) / len(ks) * TICK_US
        slabs = lambda k: (k["j"] if not k["is_x"] else k["i"] - k["j"])
        with_k = [k for k in ks if slabs(k) >= 1 and k["m"][1]]
        first = sum(k["m"][1] - k["m"][0] for k in with_k) / max(1, len(with_k)) * TICK_US
        many = [k for k in with_k if slabs(k) >= 3]
        per_slab = sum((k["m"][2] - k["m"][1]) / slabs(k) for k in many) / max(1, len(many)) * TICK_US
        print(f"  {name:10s} n {len(ks):4d}: whole {mean(lambda k: k['m'][4] - k['m'][0]):6.2f} us | start->first slab in LDS {first:5.2f} | per slab (>= 3 slabs) {per_slab:5.2f} | "
              f"K done->{'potf2 done' if name == 'L diagonal' else 'inverse tile in LDS'} {mean(lambda k: k['m'][3] - k['m'][2]):6.2f} | ->published {mean(lambda k: k['m'][4] - k['m'][3]):5.2f} | "
              f"in waits {mean(lambda k: k['wait']):6.2f}", file=out)
    # the diagonal chain: published(c) -> published(c+1), and what lies between
    diag = {k["j"]: k for k in tasks if not k["is_x"] and k["i"] == k["j"]}
    below = {k["i"]: k for k in tasks if not k["is_x"] and k["i"] == k["j"] + 1}
    steps = []
    for c in range(1, L["nb"]):
        if c in diag and c - 1 in diag and c in below:
            d0, d1, b = diag[c - 1], diag[c], below[c]
            steps.append(((d1["m"][4] - d0["m"][4]) * TICK_US,            # the step
                          (b["m"][4] - d0["m"][4]) * TICK_US,             # Linv_(c-1) published -> L(c, c-1) published
                          (d1["m"][2] - b["m"][4]) * TICK_US,             # -> diagonal tile's K loop done
                          (d1["m"][3] - d1["m"][2]) * TICK_US,            # potf2
                          (d1["m"][4] - d1["m"][3]) * TICK_US))           # stores + publish
    if steps:
        a = np.asarray(steps)
        print(f"  diagonal chain, {len(steps)} steps: published(c-1) -> published(c) {a[:, 0].mean():.2f} us = L(c,c-1) published after {a[:, 1].mean():.2f} "
              f"+ last slab of L(c,c) {a[:, 2].mean():.2f} + potf2 {a[:, 3].mean():.2f} + stores/publish {a[:, 4].mean():.2f};  sum over steps {a[:, 0].sum():.0f} us of span {span:.0f}", file=out)
        first_diag = diag.get(0)
        if first_diag: print(f"  first diagonal tile published {(first_diag['m'][4] - t0) * TICK_US:.1f} us after the first task started; last diagonal tile at "
                             f"{(diag[max(diag)]['m'][4] - t0) * TICK_US:.1f} us; span {span:.1f} us", file=out)


def run(n, mode, skip, launches, lib):
    import gpmi355x as g
    from gpmi355x import dist as gd
    rng = np.random.default_rng(20240501); d = 8
    x = rng.uniform(0.0, 1.0, size=(d, n)); y = np.sin(2 * np.pi * x).sum(axis=0) / d + 0.1 * rng.standard_normal(n)
    ll = [math.log(0.5) + 0.05 * k for k in range(d)]
    ctx = g.Context(0)
    if mode == "blocked":
        gp = gd.ShardedGPE(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1), ctx=ctx, block=1024)
    else:
        gp = g.GP(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1), ctx=ctx)
    gp.update_mll()   # (the construction may already have fitted; either way this is the warm-up)
    path = os.path.join(tempfile.gettempdir(), f"chain_trace_{os.getpid()}_{n}_{mode}.txt")
    # the trace library re-arms its launch counters when the file variable changes: set it only now
    os.environ["GPMI_CHAIN_TRACE_SKIP"] = str(skip); os.environ["GPMI_CHAIN_TRACE_LAUNCHES"] = str(launches); os.environ["GPMI_CHAIN_TRACE_FILE"] = path
    hyp = np.asarray(gp.get_params()); hyp[1:] += 1e-3
    gp.set_params(hyp); gp.update_mll()
    os.environ.pop("GPMI_CHAIN_TRACE_FILE")
    print(f"== N = {n} {mode}, launches {skip} .. {skip + launches - 1} of one update_mll!: mll {gp.mll:.9g}", flush=True)
    if os.path.exists(path):
        for L in parse(path): report(L)
        if KEEP: os.replace(path, os.path.join(KEEP, f"chain_trace_{n}_{mode}_{skip}_{launches}.txt"))
        else: os.remove(path)
    else:
        print("  (no chain launch traced)")
    del gp
    ctx.close()


def main():
    lib = os.path.join(ROOT, "tools", "bin", "libgpmi_chain_trace.so")
    if not os.path.exists(lib): raise SystemExit("run tools/build_chain_trace.sh first")
    import gpmi355x._lib as _lib
    _lib.LIB_PATH = lib
    args = sys.argv[1:]
    skip = 2; launches = 2
    cfgs = []
    global BRIEF, KEEP
    while args:
        if args[0] == "--brief": BRIEF = args[1] == "1"; args = args[2:]
        elif args[0] == "--keep": KEEP = args[1]; os.makedirs(KEEP, exist_ok=True); args = args[2:]
        elif args[0] == "--skip": skip = int(args[1]); args = args[2:]
        elif args[0] == "--launches": launches = int(args[1]); args = args[2:]
        else: cfgs.append((int(args[0]), args[1], skip, launches, BRIEF)); args = args[2:]
    for n, mode, sk, la, br in cfgs:
        BRIEF = br
        run(n, mode, sk, la, lib)


if __name__ == "__main__":
    main()
