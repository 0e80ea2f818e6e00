#!/bin/bash
O=gpurun_out/r3g
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_rccl_loopback.py "tests/test_case_wing.py::test_real_wing_on_ranks_equals_single_device" tests/test_partition_dist.py tests/test_case_bunny.py -m gpu -q --durations=8 > $O/tests.log 2>&1; tail -15 $O/tests.log
python tools/case_speed.py re266k 1000 > $O/speed_sphere.txt 2>&1; tail -2 $O/speed_sphere.txt
LUDWIG_NO_SPLIT_STEP=1 python tools/case_speed.py re266k 1000 > $O/speed_sphere_nosplit.txt 2>&1; tail -2 $O/speed_sphere_nosplit.txt
python tools/case_speed.py wing 400 > $O/speed_wing.txt 2>&1; tail -2 $O/speed_wing.txt
LUDWIG_NO_SPLIT_STEP=1 python tools/case_speed.py wing 400 > $O/speed_wing_nosplit.txt 2>&1; tail -2 $O/speed_wing_nosplit.txt
python tools/case_speed.py re10m 300 > $O/speed_re10m.txt 2>&1; tail -2 $O/speed_re10m.txt
LUDWIG_NO_SPLIT_STEP=1 python tools/case_speed.py re10m 300 > $O/speed_re10m_nosplit.txt 2>&1; tail -2 $O/speed_re10m_nosplit.txt
