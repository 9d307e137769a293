"""Step breakdown of the LAST blocked-handle step (predict of step k-1, then the fit of step k) in a rocprofv3 rocpd kernel trace of
`bench.py --mode sharded`: a fit ends at dot_kernel; the fit starts at the first covariance launch with a large grid (block-row
assembly), what lies between the previous dot_kernel and that launch is the predict.  Prints per-queue kernel sums, the all-queues-idle
time and the longest idle gaps of the main / update queues.  Usage: rocpd_blocked.py results.db"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, queue_id, grid_x, workgroup_x from kernels order by start").fetchall()
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").replace("gpmi::", "").split("(")[0][:46]
ends = [i for i, r in enumerate(rows) if "dot_kernel" in r[0]]
i1, ip = ends[-1], ends[-2]
sel = rows[ip + 1:i1 + 1]
fs = next(i for i, r in enumerate(sel) if "cov_" in r[0] and r[4] // max(r[5], 1) > 64)
pred, fit = sel[:fs], sel[fs:]
print("predict: %.2f ms, %d kernels; fit: %.2f ms, %d kernels" % ((pred[-1][2] - pred[0][1]) / 1e6, len(pred), (fit[-1][2] - fit[0][1]) / 1e6, len(fit)))


def summarize(part, title, top=12):
    by = defaultdict(lambda: [0, 0])
    for name, st, en, q, gx, wx in part:
        by[(q, short(name))][0] += 1
        by[(q, short(name))][1] += en - st
    print(title)
    for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:top]:
        print("   q%d %-46s x%-5d %8.2f ms" % (k[0], k[1], v[0], v[1] / 1e6))


summarize(pred, "predict kernels", 6)
summarize(fit, "fit kernels", 16)
# queues that carry the trailing update are the critical path: merge them, list their idle time
upd_q = {r[3] for r in fit if ("gemm_nt_kernel" in r[0] and ", 0, 4>" in r[0]) or "update256_kernel" in r[0]}
iv = sorted((r[1], r[2], short(r[0])) for r in fit if r[3] in upd_q)
t0 = fit[0][1]
ce, idle, gaps, prev = iv[0][1], 0, [], iv[0][2]
for s, e, nm in iv[1:]:
    if s > ce:
        idle += s - ce
        gaps.append((s - ce, (ce - t0) / 1e6, prev, nm))
    if e > ce:
        ce, prev = e, nm
print("update queues %s: idle %.2f ms of %.2f" % (sorted(upd_q), idle / 1e6, (fit[-1][2] - t0) / 1e6))
for g, at, a, b in sorted(gaps, reverse=True)[:12]:
    print("   gap %.2f ms at %.1f ms: after %s, before %s" % (g / 1e6, at, a, b))
