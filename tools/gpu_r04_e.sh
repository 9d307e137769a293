#!/bin/bash
# round 4, call E: the whole -m gpu suite; the blocked handle on one rank (world-1 overhead) with the lower-mode update shape
mkdir -p gpurun_out; O=gpurun_out
timeout 2700 python -m pytest tests -q -m gpu -x -s > $O/r04_e_tests.log 2>&1; grep -v amdgpu $O/r04_e_tests.log | grep -E "passed|failed|injected-latency|C4 N|Error|assert " | cut -c1-2200 | tail -12
echo "== blocked handle on one rank vs dense"
for n in 50000 20000; do
  for mode in single sharded; do
    m=""; [ $mode = sharded ] && m="--mode sharded"
    timeout 600 python bench.py --n $n --steps 5 --warmup 2 --no-cpu-baseline --no-secondary $m 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('n=$n $mode: ms %.1f fit %.1f predict %.1f frac %.3f' % (j['ms_per_step'], j['fit_only_ms_per_step'], j['predict_only_ms_per_step'], j['roofline']['frac']))"
  done
done 2>&1 | tee $O/r04_e_world1.log
