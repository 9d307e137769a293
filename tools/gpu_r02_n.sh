#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/gpurun_out/prof_sh" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --mode sharded > "$GRAFT_REPO_ROOT/gpurun_out/prof_sh.log" 2>&1
DB=$(find "$GRAFT_REPO_ROOT/gpurun_out/prof_sh" -name "*_results.db" | head -1)
python - "$DB" <<'P'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, queue_id, grid_x, workgroup_x from kernels order by start").fetchall() if True else []
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").replace("gpmi::", "").split("(")[0]
updall = [i for i, r in enumerate(rows) if "gemm_nt_kernel<double, 0, 4>" in r[0]]
i0 = updall[-48] - 40
sel = rows[i0:]
t0 = sel[0][1]
upd = [r for r in sel if "gemm_nt_kernel<double, 0, 4>" in r[0]]
print("updates:", len(upd), "total ms", sum(r[2]-r[1] for r in upd)/1e6)
for r in upd[:12]: print(f"  start {(r[1]-t0)/1e3:9.1f} us dur {(r[2]-r[1])/1e3:9.1f} us grid {r[4]//max(r[5],1)}")
from collections import defaultdict
agg = defaultdict(lambda: [0,0])
for r in sel:
    k = (short(r[0]), r[3]); agg[k][0]+=1; agg[k][1]+=r[2]-r[1]
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:16]: print(f"{k[0]:50s} q{k[1]} x{v[0]:5d} {v[1]/1e6:9.2f} ms")
print("span ms", (max(r[2] for r in sel)-t0)/1e6)
# window around update 8
u = upd[8]; w0,w1 = u[1]-3_000_000, u[1]+1_000_000
print("kernels in the 3 ms before update #8 and 1 ms after its start:")
for r in sel:
    if r[2] >= w0 and r[1] <= w1: print(f"{(r[1]-u[1])/1e3:10.1f} {(r[2]-r[1])/1e3:9.1f} q{r[3]} {r[4]//max(r[5],1) if r[4]>8192 else r[4]:6d} {short(r[0])}")
P
rm -rf "$GRAFT_REPO_ROOT/gpurun_out/prof_sh"
