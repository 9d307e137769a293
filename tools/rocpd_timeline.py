"""Timeline of the LAST fit in a rocprofv3 (rocpd SQLite) kernel trace: per stream/queue the kernels in start order with
start offset, duration and the idle gap before each; plus totals.  Usage: rocpd_timeline.py results.db [max_rows]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("# columns:", cols)
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute(f"select name, start, end, {qcol or '0'} from kernels order by start").fetchall()
# the last fit starts at the last cov_kernel launch that writes the N x N matrix: take the last cov_kernel with the longest duration
covs = [i for i, r in enumerate(rows) if "cov_kernel" in r[0] and (r[2] - r[1]) > 300000]
i0 = covs[-1]
sel = rows[i0:]
t0 = sel[0][1]
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").replace("gpmi::", "").split("(")[0]
maxrows = int(sys.argv[2]) if len(sys.argv) > 2 else 120
last_end = {}
busy = {}
print("# t_us  dur_us  gap_us  queue  kernel")
for k, (name, st, en, q) in enumerate(sel):
    gap = (st - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = en
    busy[q] = busy.get(q, 0) + (en - st)
    if k < maxrows:
        print(f"{(st - t0) / 1e3:10.1f} {(en - st) / 1e3:8.1f} {gap:8.1f}  {q}  {short(name)}")
print("# span_us", (max(r[2] for r in sel) - t0) / 1e3, "busy_us per queue", {q: v / 1e3 for q, v in busy.items()})
