#!/bin/bash
O=gpurun_out/r3h
mkdir -p $O
for i in 1 2; do
for m in 0 1 2; do
LUDWIG_SPLIT_STEP=$m python tools/case_speed.py re266k 1000 2>&1 | tail -1 | sed "s/^/sphere split mode $m: /"
done; done > $O/ab_sphere2.txt 2>&1
cat $O/ab_sphere2.txt
for m in 0 1 0 1; do
LUDWIG_SPLIT_STEP=$m python tools/case_speed.py wing 400 2>&1 | tail -1 | sed "s/^/wing split mode $m: /"
done > $O/ab_wing2.txt 2>&1
cat $O/ab_wing2.txt
for m in 0 1; do
LUDWIG_SPLIT_STEP=$m python tools/case_speed.py re10m 300 2>&1 | tail -1 | sed "s/^/re10m split mode $m: /"
done > $O/ab_re10m2.txt 2>&1
cat $O/ab_re10m2.txt
