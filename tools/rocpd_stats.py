"""Export the per-kernel summary (what `rocprofv3 --kernel-trace --stats` tabulates) from the rocpd
SQLite database rocprofv3 writes on this image, as CSV on stdout."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute(
    "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by 3 desc"
).fetchall()
tot = sum(r[2] for r in rows) or 1
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
for name, calls, total, avg, mn, mx in rows:
    print(f"\"{name}\",{calls},{total},{avg:.1f},{mn},{mx},{100.0 * total / tot:.2f}")
