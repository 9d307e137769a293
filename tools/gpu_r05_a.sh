#!/bin/bash
# round 5, call A: (1) the new parity tests (offset inputs, mean-only update on dense / blocked handles); (2) baselines before the chain
# kernel: dense bench line with c2 / c3, the blocked handle on one rank with its per-step phase times; (3) the multi-GPU line rehearsed
# on one GPU (two CU partitions, gloo).
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
echo "== new tests"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "offset or mean_only or cov_symmetric or fit_sizes" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_dist.py -q -m gpu -x -k "single_rank_device_ops or virtual" 2>&1 | tail -4
} > $O/r05_a_tests.log 2>&1
cat $O/r05_a_tests.log
{
echo "== dense bench (no CPU baseline), c2 + c3"
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary c2,c3 2>&1 | tail -1 > $O/r05_a_bench_dense.json
python - <<'EOF'
import json
j = json.load(open("gpurun_out/r05_a_bench_dense.json"))
print("dense: ms/step", j["ms_per_step"], "fit", j["fit_only_ms_per_step"], "predict", j["predict_only_ms_per_step"], "roofline", {k: j["roofline"].get(k) for k in ("frac", "peak_measured", "frac_of_measured", "avg_launch_ms")})
print("c2:", {k: j["c2"].get(k) for k in ("ms_per_step", "fit_only_ms_per_step", "predict_only_ms_per_step", "roofline_frac")})
print("c3:", {k: j["c3"].get(k) for k in ("ms_per_step", "cov_ms_per_step", "parity")})
EOF
echo "== blocked handle on one rank: N = 50 000 and c2, with per-step phases"
timeout 600 python bench.py --steps 3 --warmup 1 --mode sharded --no-cpu-baseline --secondary c2 2>&1 | tail -1 > $O/r05_a_bench_blocked.json
python - <<'EOF'
import json
j = json.load(open("gpurun_out/r05_a_bench_blocked.json"))
print("blocked one rank: ms/step", j["ms_per_step"], "fit", j["fit_only_ms_per_step"], "predict", j["predict_only_ms_per_step"], "phases", j.get("per_step_ms"))
print("c2 blocked:", {k: j["c2"].get(k) for k in ("ms_per_step", "fit_only_ms_per_step", "predict_only_ms_per_step", "per_step_ms")})
EOF
echo "== rehearsal of the multi-GPU line on one GPU"
timeout 900 python bench.py --gpus 2 --dry-run-one-gpu --n 16384 --c4-n 24576 --steps 2 --warmup 1 > $O/r05_a_rehearsal.json 2> $O/r05_a_rehearsal.err
echo "rc $?"; tail -c 3000 $O/r05_a_rehearsal.json; tail -5 $O/r05_a_rehearsal.err | cut -c1-600
} > $O/r05_a_bench.log 2>&1
cat $O/r05_a_bench.log | cut -c1-3000
