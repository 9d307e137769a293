"""GPU idle time inside the timed steps of a bench trace (rocprofv3 rocpd database): union of all kernel intervals over
all queues, per step (a step starts at a big cov_kernel launch).  Usage: rocpd_idle.py results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "cov_kernel" in r[0] and (r[2] - r[1]) > 300000]
for a, b in zip(starts[-4:-1], starts[-3:]):
    seg = rows[a:b]
    t0, t1 = seg[0][1], rows[b][1]
    busy, cur_s, cur_e = 0, None, None
    for _, s, e in sorted((r[1], r[1], r[2]) for r in seg):
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    gaps = []
    prev_e = seg[0][2]
    for _, s, e in sorted((r[1], r[1], r[2]) for r in seg):
        if s > prev_e: gaps.append(s - prev_e)
        prev_e = max(prev_e, e)
    big = sorted(gaps)[-5:]
    print(f"step: span {(t1 - t0) / 1e6:.2f} ms, GPU busy (union over queues) {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms in {len(gaps)} gaps; largest gaps (us): {[round(g / 1e3) for g in big]}")
