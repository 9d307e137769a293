#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/gpurun_out/prof_sh" -- python "$GRAFT_REPO_ROOT/bench.py" --n 20000 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --mode sharded > "$GRAFT_REPO_ROOT/gpurun_out/prof_sh.log" 2>&1
DB=$(find "$GRAFT_REPO_ROOT/gpurun_out/prof_sh" -name "*_results.db" | head -1)
python - "$DB" > "$GRAFT_REPO_ROOT/gpurun_out/sharded20k_trace.txt" <<'P'
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, queue_id, grid_x, workgroup_x from kernels order by start").fetchall()
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").replace("gpmi::", "").split("(")[0][:60]
updall = [i for i, r in enumerate(rows) if "gemm_nt_kernel<double, 0, 4>" in r[0]]
nupd = 19
i0 = updall[-nupd] - 30
sel = rows[i0:]
t0 = sel[0][1]
mainq = rows[updall[-1]][3]
print("span ms", (max(r[2] for r in sel)-t0)/1e6)
kt = defaultdict(lambda:[0,0,0,0]); prev=None
for name, st, en, q, gx, wx in sel:
    if q != mainq: continue
    k = short(name); kt[k][0]+=1; kt[k][1]+=en-st
    if prev is not None and st>prev: kt[k][2]+=1; kt[k][3]+=st-prev
    prev = en if prev is None else max(prev,en)
for k,v in sorted(kt.items(), key=lambda kv:-(kv[1][1]+kv[1][3])): print(f"{k:62s} x{v[0]:5d} {v[1]/1e6:8.2f} ms   gaps before: {v[3]/1e6:8.2f} ms")
print("timeline of the last 2.5 steps (main queue and side queue):")
u = rows[updall[-3]]
for name, st, en, q, gx, wx in sel:
    if st >= u[1]-200000: print(f"{(st-u[1])/1e3:10.1f} {(en-st)/1e3:9.1f} q{q} {gx//max(wx,1) if gx>8192 else gx:6d} {short(name)}")
P
rm -rf "$GRAFT_REPO_ROOT/gpurun_out/prof_sh"
