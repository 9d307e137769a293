#!/bin/bash
# round 6 records (what profiles/r06_* is made with): smoke, the default bench line (cpu_baseline + parity + secondary objects), kernel
# stats + critical path of the bench command and of C2, the PMC passes over the bench command (HBM traffic / MFMA busy of the trailing
# update), the blocked handle on one rank, the overlap / partition test.  The full -m gpu suite: tools/gpu_r06_d.sh.
mkdir -p gpurun_out; O=gpurun_out
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -2 | tee $O/r06_smoke.log
timeout 1200 python bench.py > $O/r06_bench.json 2> $O/r06_bench.err; echo "bench rc $?"; cut -c1-700 $O/r06_bench.json; tail -3 $O/r06_bench.err | grep -v amdgpu
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_end" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > "$GRAFT_REPO_ROOT/$O/prof_end.log" 2>&1
timeout 400 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/prof_c2" -- python "$GRAFT_REPO_ROOT/bench.py" --n 20000 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > "$GRAFT_REPO_ROOT/$O/prof_c2.log" 2>&1
cd "$GRAFT_REPO_ROOT"; DB=$(find $O/prof_end -name "*_results.db" | head -1)
python tools/rocpd_stats.py "$DB" > $O/r06_bench_kernel_stats.csv
python tools/rocpd_mainstream.py "$DB" > $O/r06_bench_critical_path.txt 2>&1
DB2=$(find $O/prof_c2 -name "*_results.db" | head -1); python tools/rocpd_mainstream.py "$DB2" > $O/r06_c2_critical_path.txt 2>&1
python tools/rocpd_stats.py "$DB2" > $O/r06_c2_kernel_stats.csv
head -6 $O/r06_bench_kernel_stats.csv | cut -c1-200; head -5 $O/r06_bench_critical_path.txt; head -5 $O/r06_c2_critical_path.txt
rm -rf $O/prof_end $O/prof_c2
bash tools/gpu_pmc_bench.sh > $O/r06_pmc_bench.log 2>&1; cp $O/bench_pmc_hbm.json $O/r06_bench_pmc_hbm.json; cut -c1-900 $O/r06_bench_pmc_hbm.json
rm -rf $O/pmc_bench
echo "== blocked handle on one rank"
timeout 600 python bench.py --steps 3 --warmup 1 --mode sharded --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 > $O/r06_blocked_world1.json
python - <<'PY' | tee $O/r06_blocked_world1.log
import json
d = json.load(open("gpurun_out/r06_bench.json")); b = json.load(open("gpurun_out/r06_blocked_world1.json"))
print("N=50000: dense %.1f ms per fit+predict, blocked handle on one rank %.1f (+%.1f %%)" % (d["ms_per_step"], b["ms_per_step"], 100 * (b["ms_per_step"] / d["ms_per_step"] - 1)))
print("N=20000: dense %.2f, blocked %.2f (+%.1f %%)" % (d["c2"]["ms_per_step"], b["c2"]["ms_per_step"], 100 * (b["c2"]["ms_per_step"] / d["c2"]["ms_per_step"] - 1)))
for k in ("per_step_ms",):
    print("N=50000 phases per block step:", {n: round(v["ms_per_block_step"], 3) for n, v in b[k].items() if isinstance(v, dict) and v.get("ms_per_block_step")})
    print("N=20000 phases per block step:", {n: round(v["ms_per_block_step"], 3) for n, v in b["c2"][k].items() if isinstance(v, dict) and v.get("ms_per_block_step")})
PY
