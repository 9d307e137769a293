#!/bin/bash
# round 5, call B: the persistent chain kernel (csrc/chain.hip): correctness first, then A/B against the multi-launch chain (GPMI_CHAIN=0)
# on the dense path (N = 50 000 and C2) and on the blocked handle on one rank; then the multi-GPU line rehearsed on one GPU.
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
echo "== chain kernel tests"
timeout 900 python -m pytest tests/test_gpu_chain.py -q -m gpu -x 2>&1 | tail -15
echo "== the suites that factor through it by default"
timeout 900 python -m pytest tests/test_gpu_twolevel.py tests/test_gpu_parity.py -q -m gpu -x -k "twolevel or width or lookahead or posdef or fit_sizes or offset or mean_only or predict_fp64" 2>&1 | tail -5
} > $O/r05_b_tests.log 2>&1
cat $O/r05_b_tests.log
summ() { python - "$1" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
except Exception as e:
    print("no line:", e); sys.exit(0)
print("  N=50000: ms/step %.1f fit %.1f predict %.1f  update frac %.3f" % (j["ms_per_step"], j["fit_only_ms_per_step"], j["predict_only_ms_per_step"], j["roofline"]["frac"]))
ph = j.get("per_step_ms")
if ph: print("   phases/step:", {k: round(v["ms_per_block_step"], 3) for k, v in ph.items() if isinstance(v, dict) and v.get("ms_per_block_step")})
c2 = j.get("c2", {})
print("  c2: ms/step %.2f fit %.2f predict %.2f update frac %.3f" % (c2.get("ms_per_step", 0), c2.get("fit_only_ms_per_step", 0), c2.get("predict_only_ms_per_step", 0), c2.get("roofline_frac", 0)))
ph = c2.get("per_step_ms")
if ph: print("   phases/step:", {k: round(v["ms_per_block_step"], 3) for k, v in ph.items() if isinstance(v, dict) and v.get("ms_per_block_step")})
PY
}
{
for CH in 1 0; do
  echo "== dense, GPMI_CHAIN=$CH"
  GPMI_CHAIN=$CH timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 > $O/r05_b_dense_chain$CH.json; summ $O/r05_b_dense_chain$CH.json
  echo "== blocked one rank, GPMI_CHAIN=$CH"
  GPMI_CHAIN=$CH timeout 600 python bench.py --steps 3 --warmup 1 --mode sharded --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 > $O/r05_b_blocked_chain$CH.json; summ $O/r05_b_blocked_chain$CH.json
done
echo "== rehearsal of the multi-GPU line on one GPU"
timeout 900 python bench.py --gpus 2 --dry-run-one-gpu --n 16384 --c4-n 24576 --steps 2 --warmup 1 > $O/r05_b_rehearsal.json 2> $O/r05_b_rehearsal.err
echo "rc $?"; tail -c 2500 $O/r05_b_rehearsal.json; tail -5 $O/r05_b_rehearsal.err | cut -c1-600
} > $O/r05_b_bench.log 2>&1
cat $O/r05_b_bench.log | cut -c1-2500
