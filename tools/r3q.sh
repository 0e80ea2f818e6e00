#!/bin/bash
O=gpurun_out/r3q
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_case_bunny.py tests/test_golden.py tests/test_case_wing.py::test_wing_hip_equals_oracle tests/test_case_ball1m.py::test_cube1m_hip_equals_oracle -m gpu -q -x > $O/tests.log 2>&1; tail -4 $O/tests.log
for i in 1 2; do
for m in fused copy; do
  if [ $m = copy ]; then export LUDWIG_RHO_OLD_COPY=1; else unset LUDWIG_RHO_OLD_COPY; fi
  python tools/case_speed.py re266k 1000 2>&1 | tail -1 | sed "s/^/sphere rho_old $m: /"
  python tools/case_speed.py wing 400 2>&1 | tail -1 | sed "s/^/wing   rho_old $m: /"
  python tools/case_speed.py re10m 300 2>&1 | tail -1 | sed "s/^/re10m  rho_old $m: /"
done; done > $O/ab_rho_old.txt 2>&1
unset LUDWIG_RHO_OLD_COPY
cat $O/ab_rho_old.txt
