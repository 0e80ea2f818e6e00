#!/bin/bash
O=gpurun_out/r3q
mkdir -p $O
for i in 1 2 3; do
for m in 1 2; do
  export LUDWIG_LEVEL_STREAM_PRIORITY=$m
  python tools/case_speed.py re266k 1000 2>&1 | tail -1 | sed "s/^/sphere priorities mode $m: /"
  python tools/case_speed.py wing 400 2>&1 | tail -1 | sed "s/^/wing   priorities mode $m: /"
  python tools/case_speed.py re10m 300 2>&1 | tail -1 | sed "s/^/re10m  priorities mode $m: /"
done; done > $O/ab_prio2.txt 2>&1
cat $O/ab_prio2.txt
