#!/bin/bash
# First GPU shake-down: per-group pytest (separate processes so a device fault only loses one group),
# MFMA peak micro-benchmark, a short bench, and a rocprofv3 kernel trace of the bench.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
python - <<'PY' > gpurun_out/peak.log 2>&1
import sys; sys.path.insert(0, "gaussianprocesses.jl_amd")
import gpmi355x as g
c = g.Context.default(0)
for bits in (64, 32):
    print("mfma peak", bits, [round(c.mfma_peak(bits), 2) for _ in range(3)], "TFLOP/s")
PY
cat gpurun_out/peak.log
for grp in "cov" "fit or factor or pdmat or means or hetero or posdef or constructor" "predict or one_dim or fp32 and not cov" "golden or synthetic or optimize"; do
  name=$(echo "$grp" | tr ' ' '_' | cut -c1-20)
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "$grp" -p no:cacheprovider > "gpurun_out/pytest_$name.log" 2>&1
  echo "== pytest [$grp] exit $?"; tail -n 25 "gpurun_out/pytest_$name.log"
done
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench1.log 2>&1; echo "bench exit $?"; tail -n 5 gpurun_out/bench1.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof1" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof1.log" 2>&1; echo "rocprof exit $?"
cd "$GRAFT_REPO_ROOT"; find gpurun_out/prof1 -name "*stats*" | head; f=$(find gpurun_out/prof1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f"
