#!/bin/bash
# round 5, call K: the driver's invocation run by hand at the final tree, and smoke()
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_driver_like.json 2> $O/r05_bench_driver_like.err ) 2>&1 | tail -3
echo "rc $?"; cut -c1-400 $O/r05_bench_driver_like.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
