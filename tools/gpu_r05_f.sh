#!/bin/bash
# round 5, call F: predict_f on two streams A/B (the GPMI_PREDICT_SPLIT experiment: no gain, removed again afterwards together with the
# "predict" mode of tools/knob_sweep.py — profiles/r05_f_predict_split.log), then the whole GPU suite as the driver runs it
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
echo "== predict split A/B"
for N in 20000 50000; do timeout 600 python tools/knob_sweep.py $N predict 2>&1 | grep "^N="; done
} > $O/r05_f_predict_split.log 2>&1
cat $O/r05_f_predict_split.log
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 > $O/r05_f_gpu_tests.log
cat $O/r05_f_gpu_tests.log
