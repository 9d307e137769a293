#!/bin/bash
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/gpurun_out/prof_lib" -- python "$GRAFT_REPO_ROOT/tools/yardstick_k256.py" > "$GRAFT_REPO_ROOT/gpurun_out/prof_lib.log" 2>&1
cd "$GRAFT_REPO_ROOT"; DB=$(find gpurun_out/prof_lib -name "*_results.db" | head -1)
python - "$DB" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for name, n, avg, gx, wx, lds, vg, ag in db.execute("select name, count(*), avg(duration), max(grid_x), max(workgroup_x), max(lds_size), max(vgpr_count), max(accum_vgpr_count) from kernels group by name order by sum(duration) desc limit 6"):
    print(f"{n:4d} x {avg/1e3:9.1f} us  grid {gx} wg {wx} lds {lds} vgpr {vg} agpr {ag}  {name[:260]}")
PY
rm -rf gpurun_out/prof_lib
