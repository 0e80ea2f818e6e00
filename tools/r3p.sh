#!/bin/bash
O=gpurun_out/r3p
mkdir -p $O
export TMPDIR=/tmp
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
( time timeout -k 10 1000 python -m pytest tests -m gpu -q -x --durations=12 ) > $O/gpu_suite.log 2>&1; tail -22 $O/gpu_suite.log
python tools/case_speed.py re266k 1000 2>&1 | tail -1
python tools/case_speed.py wing 400 2>&1 | tail -1
python tools/case_speed.py re10m 300 2>&1 | tail -1
