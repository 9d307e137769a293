#!/bin/bash
# round 5, call G: is the sharded fit over two CU partitions slower with the chain kernel (2186 ms in the suite run against 1953 - 1993 in round 4),
# or is it the box?  A/B on one box.  Then: a narrower FIRST super-panel.
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
for CH in 1 0 1; do
  echo "== overlap test, GPMI_CHAIN=$CH"
  GPMI_CHAIN=$CH timeout 900 python -m pytest tests/test_gpu_dist.py -q -m gpu -s -k "cu_partitions_and_injected" 2>&1 | grep -E "injected-latency|passed|failed" | cut -c1-900
done
echo "== first panel width"
for N in 20000 50000; do timeout 600 python tools/knob_sweep.py $N first 2>&1 | grep "^N="; done
} > $O/r05_g_partitions_ab.log 2>&1
cat $O/r05_g_partitions_ab.log
