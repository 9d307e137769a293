#!/bin/bash
# round 5, call E: the whole GPU suite as the driver runs it, at the tree with the chain kernel, the fused tail and the new thresholds
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 > $O/r05_e_gpu_tests.log
cat $O/r05_e_gpu_tests.log
ls gpurun_out/*.json 2>/dev/null | head
