#!/bin/bash
# block-major library: headline bench, rate against block count, sphere / wing stepping rates - one box
O=gpurun_out/r3d
mkdir -p $O
python bench.py --steps 100 --warmup 10 --cpu-seconds 0 > $O/bench256.json 2> $O/bench256.err && cat $O/bench256.json &&
python bench.py > $O/bench_default.json 2>> $O/bench256.err && cat $O/bench_default.json &&
python tools/stride_padding.py > $O/rate_vs_block_count.txt 2>&1 && cat $O/rate_vs_block_count.txt &&
python bench.py --size 512 --steps 30 --warmup 5 --cpu-seconds 0 > $O/bench512.json 2>> $O/bench256.err && cat $O/bench512.json &&
python tools/case_speed.py > $O/case_speed.txt 2>&1; tail -20 $O/case_speed.txt
