#!/bin/bash
O=gpurun_out/r3l
mkdir -p $O
export TMPDIR=/tmp
( time timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=15 ) > $O/gpu_suite.log 2>&1; tail -30 $O/gpu_suite.log
python tools/run_wing.py $O/r03_wing5deg_10000_steps 10000 100 > $O/run_wing.log 2>&1; tail -40 $O/run_wing.log
