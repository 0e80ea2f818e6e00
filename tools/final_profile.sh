#!/bin/bash
# Round-end evidence, all on ONE box: the bench line, the rocprofv3 kernel-trace of the same command, the two PMC passes, and the
# data-movement floor of the box (tools/stridebench: the step's bytes in the product's schedule, block-major as the library stores them;
# tools/marchbench: round 2's population-major floors) so that the kernel time can be read against what this very device delivers.
set -e
export TMPDIR=/tmp
R=${ROUND:-r03}
O=gpurun_out/final
rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
{ tools/stridebench 32 20 3 | tail -3; tools/stridebench 32 20 0 32768 32768 1 | tail -1 | sed 's/^/population-major at 64 MiB: /'; tools/marchbench 32 9 47; } > $O/floor.txt 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 bench.py --steps 100 --warmup 10 --cpu-seconds 0 > $O/bench_under_rocprof.json 2> $O/trace.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 tools/profile_step.py 256 12 > $O/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python3 tools/profile_step.py 256 12 > $O/pmc_write.log 2>&1
python3 tools/summarize_profile.py $O/trace $O $O/${R}_bench256 16777216 > $O/summary.txt
cp $O/trace/t_kernel_stats.csv $O/${R}_bench256_rocprofv3_kernel_stats.csv
rm -rf $O/trace/t_kernel_trace.csv $O/pmc_fetch/*kernel_trace.csv $O/pmc_write/*kernel_trace.csv
python bench.py --steps 100 --warmup 10 --cpu-seconds 0 > $O/bench_again.json 2>> $O/bench.err
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_driver_args.json 2>> $O/bench.err
LUDWIG_REFERENCE_BLOCK_ORDER=1 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 > $O/bench_reference_block_order.json 2>> $O/bench.err
python bench.py --size 512 --steps 30 --warmup 5 --cpu-seconds 0 > $O/bench512.json 2>> $O/bench.err
python bench.py --scaling strong --steps 30 --warmup 5 --cpu-seconds 0 > $O/bench_strong_n1.json 2>> $O/bench.err
LUDWIG_BENCH_FORCE_DEVICE=0 python bench.py --gpus 2 --size 128 --steps 20 --warmup 5 > $O/bench_2rank_rehearsal_weak.json 2>> $O/bench.err
LUDWIG_BENCH_FORCE_DEVICE=0 python bench.py --gpus 2 --scaling strong --size 256 --steps 20 --warmup 5 > $O/bench_2rank_rehearsal_strong.json 2>> $O/bench.err
cat $O/bench.json; cat $O/bench_under_rocprof.json; cat $O/floor.txt; cat $O/summary.txt
