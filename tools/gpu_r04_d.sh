#!/bin/bash
# round 4, call D: the whole -m gpu suite on the tree with the new covariance / gradient kernels, the run-time-sized kernel program, CU
# partitions and the PDMat surface on blocked handles; then cov timings with the cheaper Noise prefilter
mkdir -p gpurun_out; O=gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x -s > $O/r04_d_tests.log 2>&1; grep -v amdgpu $O/r04_d_tests.log | grep -E "passed|failed|injected-latency|CU partitions|C4 N|Error|assert " | cut -c1-1800 | tail -12
cd /tmp; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
for w in c3; do
  P="$R/$O/pmc_cov_$w"; rm -rf "$P"; mkdir -p "$P"
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/kt" -- python "$R/tools/cov_only.py" $w > "$P/kt.log" 2>&1
  grep -h "cov_\|scale_inputs" $(find "$P/kt" -name "*kernel_stats.csv") | cut -c1-60,170-260
  rm -rf "$P"
done
