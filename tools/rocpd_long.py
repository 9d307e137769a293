"""Kernels longer than a threshold, in start order (rocprofv3 rocpd database).  Usage: rocpd_long.py results.db [min_us]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
thr = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 5e6
rows = db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
t0 = rows[0][1]
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").replace("gpmi::", "").split("(")[0][:48]
for name, st, en, gx, wx in rows:
    if en - st >= thr:
        print(f"{(st - t0) / 1e6:10.2f} ms  {(en - st) / 1e6:9.3f} ms  wgs {gx // max(wx, 1):7d}  {short(name)}")
