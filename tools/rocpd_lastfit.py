"""Every kernel of the LAST fit + predict in a rocprofv3 rocpd kernel trace, all queues, in start order:
t_us (from the fit's covariance launch), duration, idle gap before it on ITS queue, queue, workgroups, kernel.
Usage: rocpd_lastfit.py results.db [max_rows]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = db.execute(f"select name, start, end, {qcol}, grid_x, workgroup_x from kernels order by start").fetchall()
fin = [i for i, r in enumerate(rows) if "finalize" in r[0]]
i1 = fin[-1]
lo = fin[-2] if len(fin) > 1 else 0
i0 = max(i for i in range(lo, i1) if "cov_" in rows[i][0] and rows[i][4] // max(rows[i][5], 1) > 256)
sel = rows[i0:]
t0 = sel[0][1]
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").replace("gpmi::", "").split("(")[0]
maxrows = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
last_end = {}
print("# t_us  dur_us  gap_us  queue  wgs  kernel")
for k, (name, st, en, q, gx, wx) in enumerate(sel[:maxrows]):
    gap = (st - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = max(en, last_end.get(q, 0))
    mark = "  <- end of fit" if i0 + k == i1 else ""
    print(f"{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f} {gap:7.1f}  {q}  {gx // max(wx, 1):5d}  {short(name)}{mark}")
