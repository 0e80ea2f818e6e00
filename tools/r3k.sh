#!/bin/bash
O=gpurun_out/r3k
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_case_bunny.py tests/test_golden.py -m gpu -q -x > $O/tests.log 2>&1; tail -4 $O/tests.log
for i in 1 2; do
for m in 0 1; do
  if [ $m = 1 ]; then export LUDWIG_IFACE_TWO_KERNELS=1; else unset LUDWIG_IFACE_TWO_KERNELS; fi
  python tools/case_speed.py re266k 1000 2>&1 | tail -1 | sed "s/^/sphere two_kernels=$m: /"
  python tools/case_speed.py wing 400 2>&1 | tail -1 | sed "s/^/wing   two_kernels=$m: /"
  python tools/case_speed.py re10m 300 2>&1 | tail -1 | sed "s/^/re10m  two_kernels=$m: /"
done; done > $O/ab_fused.txt 2>&1
unset LUDWIG_IFACE_TWO_KERNELS
cat $O/ab_fused.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt_wing -o t --output-format csv -- python3 $R/tools/case_speed.py wing 200 > $O/wing_prof.log 2>&1; head -8 $O/kt_wing/t_kernel_stats.csv | cut -c1-170; rm -f $O/kt_wing/t_kernel_trace.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt_sphere -o t --output-format csv -- python3 $R/tools/case_speed.py re266k 400 > $O/sphere_prof.log 2>&1; head -8 $O/kt_sphere/t_kernel_stats.csv | cut -c1-170; rm -f $O/kt_sphere/t_kernel_trace.csv
