#!/bin/bash
O=gpurun_out/r3m
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_rccl_loopback.py "tests/test_case_wing.py::test_real_wing_on_ranks_equals_single_device" tests/test_case_ball1m.py::test_ball1m_on_two_ranks_equals_single_device tests/test_gpu_parity.py::test_tunnel_sphere_bit_exact -m gpu -q --durations=6 > $O/tests.log 2>&1; tail -25 $O/tests.log
