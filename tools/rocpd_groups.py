"""Per (kernel, grid size class) totals from a rocprofv3 rocpd database: separates the GEMM launches by role.
Usage: rocpd_groups.py results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, grid_x, workgroup_x, duration from kernels").fetchall()
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").replace("gpmi::", "").split("(")[0][:60]
agg = {}
for name, gx, wx, dur in rows:
    wgs = gx // max(wx, 1)
    cls = "<=16" if wgs <= 16 else "<=64" if wgs <= 64 else "<=256" if wgs <= 256 else "<512" if wgs < 512 else "512+"
    k = (short(name), cls)
    a = agg.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += dur
tot = sum(v[1] for v in agg.values())
print(f"{'kernel':62s} {'WGs':6s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'%':>6s}")
for (n, cls), (cnt, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:62s} {cls:6s} {cnt:7d} {d / 1e6:10.3f} {d / cnt / 1e3:9.1f} {100 * d / tot:6.2f}")
