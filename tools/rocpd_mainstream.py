"""Critical-path breakdown of the LAST fit (cov -> factorisation -> solves) in a rocprofv3 rocpd kernel trace: the main queue
(the one with the most busy time) is walked in start order; its wall time is split into kernel time by kernel name / grid
bucket and into idle gaps attributed to the kernel that FOLLOWS each gap (waits for the side stream, launch latency).
Usage: rocpd_mainstream.py results.db"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
gcol = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else ("grid_size" if "grid_size" in cols else "0"))
wcol = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else ("workgroup_size" if "workgroup_size" in cols else "1"))
rows = db.execute(f"select name, start, end, {qcol}, {gcol}, {wcol} from kernels order by start").fetchall()
# the last fit: it ends at the last finalize kernel (dense path) / dot kernel (blocked path) and starts at the first covariance
# launch after the end marker before that (the blocked path assembles block-row by block-row: many launches)
ends = [i for i, r in enumerate(rows) if "finalize" in r[0] or "dot_kernel" in r[0]]
i1 = ends[-1]
prev_end = ends[-2] if len(ends) > 1 else -1
i0 = next(i for i in range(prev_end + 1, i1) if "cov_" in rows[i][0] and (rows[i][2] - rows[i][1]) > 20000 and not any(
    "row_gemv" in rows[j][0] or "row_sumsq" in rows[j][0] for j in range(i, min(i + 400, i1)) if rows[j][1] < rows[i][2] + 2_000_000))
sel = rows[i0:i1 + 1]
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").replace("gpmi::", "").split("(")[0]
busy = defaultdict(int)
for name, st, en, q, gx, wx in sel:
    busy[q] += en - st
# round 3: the update that hides the look-ahead chain runs on a CU-masked stream of its own (event hop from / to the main
# stream): the critical path is the UNION of the queue that carries cov / finalize and the queue with the most trailing-update
# time; every other queue is a side queue
upd_busy = defaultdict(int)
for name, st, en, q, gx, wx in sel:
    if ("gemm_nt_kernel<" in name and ", 0, 4>" in name) or "update256_kernel" in name:
        upd_busy[q] += en - st
mains = {sel[0][3], sel[-1][3]}
if upd_busy:
    mains.add(max(upd_busy, key=upd_busy.get))
main = sorted(mains)
t0, t1 = sel[0][1], sel[-1][2]
print(f"# last fit: span {(t1 - t0) / 1e6:.2f} ms; queues busy (ms): " + ", ".join(f"{q}: {v / 1e6:.1f}" for q, v in busy.items()) + f"; main = {main}")


def bucket(gx, wx):
    wgs = gx // max(wx, 1) if gx > 8192 else gx  # grid given in threads or in workgroups, depending on the rocprof version
    for b in (16, 64, 256, 511):
        if wgs <= b:
            return f"<={b}"
    return "512+"


kt = defaultdict(lambda: [0, 0])
gt = defaultdict(lambda: [0, 0])
prev_end = None
for name, st, en, q, gx, wx in sel:
    if q not in mains:
        continue
    key = f"{short(name)} [{bucket(gx, wx)}]"
    kt[key][0] += 1
    kt[key][1] += en - st
    if prev_end is not None and st > prev_end:
        gt[key][0] += 1
        gt[key][1] += st - prev_end
    prev_end = en if prev_end is None else max(prev_end, en)
tot_k = sum(v[1] for v in kt.values())
tot_g = sum(v[1] for v in gt.values())
print(f"# main queue: kernels {tot_k / 1e6:.2f} ms, idle gaps {tot_g / 1e6:.2f} ms")
print("# kernel [grid bucket]                                   calls   kernel_ms   gaps_before   gap_ms")
for key in sorted(kt, key=lambda k: -(kt[k][1] + gt[k][1])):
    print(f"{key:58s} {kt[key][0]:6d} {kt[key][1] / 1e6:10.2f} {gt[key][0]:10d} {gt[key][1] / 1e6:10.2f}")
# side queues
for q in busy:
    if q in mains:
        continue
    ks = defaultdict(lambda: [0, 0])
    for name, st, en, qq, gx, wx in sel:
        if qq == q:
            ks[short(name)][0] += 1
            ks[short(name)][1] += en - st
    print(f"# queue {q}: " + "; ".join(f"{k} x{v[0]} {v[1] / 1e6:.1f} ms" for k, v in sorted(ks.items(), key=lambda kv: -kv[1][1])))

# optional: timeline window around the k-th trailing-update launch (argv[2] = k): every kernel of every queue that overlaps
# [update start, update end + 5 ms], offsets in us relative to the update's start
if len(sys.argv) > 2:
    k = int(sys.argv[2])
    upd = [r for r in sel if "gemm_nt_kernel<double, 0, 4>" in r[0] or "gemm_nt_kernel<float, 0, 4>" in r[0] or "update256_kernel" in r[0]]
    u = upd[k]
    w0, w1 = u[1], u[2] + 5_000_000
    print(f"# window: update #{k} lasts {(u[2] - u[1]) / 1e3:.0f} us; kernels overlapping [0, {(w1 - w0) / 1e3:.0f}] us")
    print("#   start_us     end_us   dur_us  queue  grid  kernel")
    win = [r for r in sel if r[2] >= w0 and r[1] <= w1]
    show = win if len(win) <= 140 else win[:70] + [None] + win[-70:]
    for r in show:
        if r is None:
            print("   ...")
            continue
        name, st, en, q, gx, wx = r
        print(f"{(st - w0) / 1e3:11.1f} {(en - w0) / 1e3:10.1f} {(en - st) / 1e3:8.1f}  {q}  {gx // max(wx, 1) if gx > 8192 else gx:5d}  {short(name)}")
