"""Discrete-event model of the persistent chain kernel's task queue (csrc/chain.hip): W workgroups take tasks from one counter in list
order, a task multiplies its K slabs as their input tiles are published, then waits for the diagonal tile it needs.  Stage costs are the
means of the in-kernel timeline (tools/chain_trace.py, profiles/r05_s*_chain_trace*.log) — the model exists to compare TASK ORDERS on the
CPU before spending GPU minutes on them, not to predict absolute times.

Usage: python tools/chain_sim.py [nb]      prints the span of each candidate order for 8 / 16 / 32 workgroups."""
import heapq, sys

# stage costs in us (8 workgroups, one per compute unit, beside the trailing update)
T_PICK = 1.3      # task handed out -> first loads issued
T_SLAB = 2.9      # one 64-deep slab: LDS publish + 64 MFMAs per wave + the next slab's loads
T_LAT = 2.0       # a tile published by another workgroup -> in this one's LDS (flag poll + fetch), when not prefetched
T_TAIL = 3.0      # product with the inverse tile + store + write-through drain + flag
T_POTF2 = 21.5    # the one-wavefront 64-column potf2 + inverse
T_PUB_DIAG = 1.6  # L_cc, Linv_c, X_cc stores + flag


def order_round5_first(nb):
    """step c: L(c..nb-1, c), X(c, 0..c-1)"""
    out = []
    for c in range(nb):
        out += [("L", i, c) for i in range(c, nb)] + [("X", c, j) for j in range(c)]
    return out


def order_diag_ahead(nb, x_first=False):
    """L(0,0); step c: L(c+1,c), L(c+1,c+1), L(c+2.., c), X(c, .)  (chain_order.h)"""
    out = [("L", 0, 0)]
    for c in range(nb):
        if c + 1 < nb: out += [("L", c + 1, c), ("L", c + 1, c + 1)]
        bulk = [("L", i, c) for i in range(c + 2, nb)]
        xs = [("X", c, j) for j in range(c)]
        out += (xs + bulk) if x_first else (bulk + xs)
    return out


def order_row_ahead(nb, ahead=2):
    """like diag_ahead, but the tiles of the next `ahead` rows of column c come before the rest of the bulk is interleaved with X"""
    out = [("L", 0, 0)]
    for c in range(nb):
        if c + 1 < nb: out += [("L", c + 1, c), ("L", c + 1, c + 1)]
        bulk = [("L", i, c) for i in range(c + 2, nb)]
        xs = [("X", c, j) for j in range(c)]
        near, far = bulk[:ahead], bulk[ahead:]
        out += near
        # interleave the far bulk with the X tiles (both need nothing newer than L(c, c))
        k = 0
        while far or xs:
            if far: out.append(far.pop(0))
            if xs: out.append(xs.pop(0))
    return out


def check_topological(order, nb):
    pos = {t: n for n, t in enumerate(order)}
    for n, (kind, i, j) in enumerate(order):
        deps = []
        if kind == "L":
            deps += [("L", i, k) for k in range(j)] + [("L", j, k) for k in range(j)]
            if i > j: deps.append(("L", j, j))
        else:
            deps += [("L", i, k) for k in range(j, i)] + [("X", k, j) for k in range(j + 1, i)] + [("L", j, j), ("L", i, i)]
        for d in deps:
            if pos[d] >= n: return False
    return len(order) == nb * nb


def simulate(order, nb, wgs, potf2=T_POTF2, slab=T_SLAB):
    done = {}
    free = [(0.0, w) for w in range(wgs)]
    heapq.heapify(free)
    # tasks are handed out in list order to whichever workgroup asks first; a workgroup asks when it has published its previous task.
    # The dependencies of a task precede it in the list but may still be RUNNING: their completion times are known only after they are
    # simulated, and they were handed out earlier, so simulating in hand-out order is exact.
    for task in order:
        t, w = heapq.heappop(free)
        kind, i, j = task
        cur = t + T_PICK
        if kind == "L":
            for k in range(j):
                ready = max(done[("L", i, k)], done[("L", j, k)])
                cur = max(cur, ready + T_LAT) + slab
            if i == j:
                cur += potf2 + T_PUB_DIAG
            else:
                cur = max(cur + 0.7, done[("L", j, j)] + T_LAT + 0.7) + T_TAIL
        else:
            for k in range(j, i):
                ready = max(done[("L", i, k)], done[("L", j, j)] if k == j else done[("X", k, j)])
                cur = max(cur, ready + T_LAT) + slab
            cur = max(cur + 0.7, done[("L", i, i)] + T_LAT + 0.7) + T_TAIL
        done[task] = cur
        heapq.heappush(free, (cur, w))
    return max(done.values())


if __name__ == "__main__":
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    cands = {"round-5 first order (column by column)": order_round5_first(nb), "diagonal ahead (chain_order.h)": order_diag_ahead(nb),
             "diagonal ahead, X before the bulk": order_diag_ahead(nb, True)}
    for a in (1, 2, 3, 4):
        cands[f"diagonal ahead, {a} near rows first, rest interleaved with X"] = order_row_ahead(nb, a)
    for name, o in cands.items():
        assert check_topological(o, nb), name
        print(f"{name:62s}: " + "  ".join(f"{w:2d} wgs {simulate(o, nb, w, potf2=(29.0 if w == 16 else T_POTF2), slab=(4.2 if w == 16 else T_SLAB)):7.1f} us" for w in (8, 16, 32)))


# ---- experiment: the hand-out sequence of a DYNAMIC list scheduler (highest bottom level first among the tasks whose inputs are all
# ---- published or running) recorded as a static order
def deps_of(task):
    kind, i, j = task
    if kind == "L":
        d = [("L", i, k) for k in range(j)] + [("L", j, k) for k in range(j)]
        if i > j: d.append(("L", j, j))
        return d
    return [("L", i, k) for k in range(j, i)] + [("X", k, j) for k in range(j + 1, i)] + [("L", j, j), ("L", i, i)]


def cost_of(task):
    kind, i, j = task
    if kind == "L":
        return T_PICK + j * T_SLAB + (T_POTF2 + T_PUB_DIAG if i == j else 0.7 + T_TAIL)
    return T_PICK + (i - j) * T_SLAB + 0.7 + T_TAIL


def bottom_levels(nb):
    tasks = order_round5_first(nb)
    succ = {t: [] for t in tasks}
    for t in tasks:
        for d in deps_of(t): succ[d].append(t)
    bl = {}
    for t in reversed(tasks):  # reverse topological
        bl[t] = cost_of(t) + max((bl[s] for s in succ[t]), default=0.0)
    return bl


def order_by_priority(nb, lookahead_slack):
    """static order: repeatedly take the highest-bottom-level task all of whose dependencies are already in the list AND whose
    dependencies were listed at least `lookahead_slack` positions earlier unless nothing else is available"""
    bl = bottom_levels(nb)
    tasks = set(bl)
    pos, out = {}, []
    while tasks:
        avail = [t for t in tasks if all(d in pos for d in deps_of(t))]
        relaxed = [t for t in avail if all(len(out) - pos[d] >= lookahead_slack for d in deps_of(t))]
        pick = max(relaxed or avail, key=lambda t: bl[t])
        pos[pick] = len(out); out.append(pick); tasks.remove(pick)
    return out


if __name__ == "__main__":
    for slack in (0, 2, 4, 8, 12, 16, 24):
        o = order_by_priority(nb, slack)
        assert check_topological(o, nb)
        print(f"{'bottom-level priority, slack ' + str(slack):62s}: " + "  ".join(f"{w:2d} wgs {simulate(o, nb, w, potf2=(29.0 if w == 16 else T_POTF2), slab=(4.2 if w == 16 else T_SLAB)):7.1f} us" for w in (8, 16, 32)))
