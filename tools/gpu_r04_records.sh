#!/bin/bash
# round 4 records (what profiles/r04_* was made with): smoke, the default bench line (cpu_baseline + parity + secondary objects with their
# parity fields), kernel stats + critical path of the bench command, the PMC passes over the bench command (HBM traffic / MFMA busy of the
# trailing update), timings and instruction counters of the covariance kernels.  The full -m gpu suite is tools/gpu_r04_n.sh.
mkdir -p gpurun_out; O=gpurun_out
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -2 | tee $O/r04_smoke.log
timeout 1200 python bench.py > $O/r04_bench.json 2> $O/r04_bench.err; echo "bench rc $?"; cut -c1-700 $O/r04_bench.json; tail -3 $O/r04_bench.err | grep -v amdgpu
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_end" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > "$GRAFT_REPO_ROOT/$O/prof_end.log" 2>&1
timeout 400 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/prof_c2" -- python "$GRAFT_REPO_ROOT/bench.py" --n 20000 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > "$GRAFT_REPO_ROOT/$O/prof_c2.log" 2>&1
cd "$GRAFT_REPO_ROOT"; DB=$(find $O/prof_end -name "*_results.db" | head -1)
python tools/rocpd_stats.py "$DB" > $O/r04_bench_kernel_stats.csv
python tools/rocpd_mainstream.py "$DB" > $O/r04_bench_critical_path.txt 2>&1
DB2=$(find $O/prof_c2 -name "*_results.db" | head -1); python tools/rocpd_mainstream.py "$DB2" > $O/r04_c2_critical_path.txt 2>&1
head -6 $O/r04_bench_kernel_stats.csv | cut -c1-200; head -5 $O/r04_bench_critical_path.txt; head -5 $O/r04_c2_critical_path.txt
rm -rf $O/prof_end $O/prof_c2
bash tools/gpu_pmc_bench.sh > $O/r04_pmc_bench.log 2>&1; cp $O/bench_pmc_hbm.json $O/r04_bench_pmc_hbm.json; cut -c1-900 $O/r04_bench_pmc_hbm.json
rm -rf $O/pmc_bench
cd /tmp; R="$GRAFT_REPO_ROOT"
for w in seard c3 f32d16; do
  P="$R/$O/pmc_cov_$w"; rm -rf "$P"; mkdir -p "$P"
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/kt" -- python "$R/tools/cov_only.py" $w > "$P/kt.log" 2>&1
  grep -h "cov_\|scale_inputs" $(find "$P/kt" -name "*kernel_stats.csv") | cut -c1-70,170-260 | tee "$R/$O/r04_cov_stats_$w.csv"
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$P/sq1" -- python "$R/tools/cov_only.py" $w > "$P/sq1.log" 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$P/sq2" -- python "$R/tools/cov_only.py" $w > "$P/sq2.log" 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES WRITE_SIZE --output-format csv -d "$P/wr" -- python "$R/tools/cov_only.py" $w > "$P/wr.log" 2>&1
  python "$R/tools/pmc_cov_valu.py" "$P" > "$R/$O/r04_cov_pmc_$w.json" 2>&1
  rm -rf "$P"
done
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_fitc" -- python "$R/tools/fitc_bench.py" 1000000x4096 > "$R/$O/r04_fitc_c5.log" 2>&1
grep -v amdgpu "$R/$O/r04_fitc_c5.log" | grep "N=" ; (head -1 $(find "$R/$O/prof_fitc" -name "*kernel_stats.csv"); grep -h "gpmi" $(find "$R/$O/prof_fitc" -name "*kernel_stats.csv") | head -16) | cut -c1-90,150-240 | tee "$R/$O/r04_fitc_c5_kernel_stats.csv"
rm -rf "$R/$O/prof_fitc"
