#!/bin/bash
O=gpurun_out/r3r
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt_sphere -o t --output-format csv -- python3 $R/tools/case_speed.py re266k 300 > $R/$O/sphere_prof.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt_wing -o t --output-format csv -- python3 $R/tools/case_speed.py wing 150 > $R/$O/wing_prof.log 2>&1
cd $R
python3 tools/trace_timeline.py $O/kt_sphere/t_kernel_trace.csv 90 30 > $O/sphere_timeline.txt
python3 tools/trace_timeline.py $O/kt_wing/t_kernel_trace.csv 90 30 > $O/wing_timeline.txt
head -8 $O/kt_sphere/t_kernel_stats.csv | cut -c1-150; head -8 $O/kt_wing/t_kernel_stats.csv | cut -c1-150
rm -f $O/kt_sphere/t_kernel_trace.csv $O/kt_wing/t_kernel_trace.csv
cat $O/sphere_timeline.txt
