#!/bin/bash
# round 3, call Q: kernel trace of the C2 workload (N = 20000) at the final tree: critical path, groups, full timeline of the last fit + predict
mkdir -p gpurun_out; O=gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_q" -- python "$GRAFT_REPO_ROOT/bench.py" --n 20000 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > "$GRAFT_REPO_ROOT/$O/q_prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"; DB=$(find $O/prof_q -name "*_results.db" | head -1)
python tools/rocpd_mainstream.py "$DB" > $O/q_c2_critical_path.txt 2>&1
python tools/rocpd_groups.py "$DB" > $O/q_c2_groups.txt 2>&1
python tools/rocpd_lastfit.py "$DB" > $O/q_c2_lastfit.txt 2>&1
head -12 $O/q_c2_critical_path.txt; wc -l $O/q_c2_lastfit.txt
rm -rf $O/prof_q
timeout 200 python bench.py --n 20000 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print({k: j[k] for k in ('ms_per_step','fit_only_ms_per_step','predict_only_ms_per_step')}, j['stage_ms_per_step'])"
