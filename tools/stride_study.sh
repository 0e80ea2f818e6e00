#!/bin/bash
# Round 3: where are the bad population strides? (tools/stridebench.hip emulation + the real kernel at a few points), one box.
O=gpurun_out/r3a
mkdir -p $O
tools/stridebench 32 10 0 32768 49152 32 > $O/sb_nb32_coarse.txt 2>&1
tools/stridebench 32 10 0 34688 34944 1 > $O/sb_nb32_fine68.txt 2>&1
tools/stridebench 32 10 0 32768 33792 4 > $O/sb_nb32_fine64_66.txt 2>&1
tools/stridebench 32 10 1 32768 49152 128 > $O/sb_nb32_loads.txt 2>&1
tools/stridebench 32 10 2 32768 49152 128 > $O/sb_nb32_stores.txt 2>&1
tools/stridebench 24 20 0 13824 20736 16 > $O/sb_nb24.txt 2>&1
tools/stridebench 16 40 0 4096 8192 8 > $O/sb_nb16.txt 2>&1
tools/stridebench 32 10 0 32768 32768 1 884736 > /dev/null 2>&1
for g in $(seq 884736 128 893000); do tools/stridebench 32 10 0 32768 32768 1 $g | tail -1 | sed "s/^/gap $g: /"; done > $O/sb_nb32_gap.txt 2>&1
python tools/stride_padding.py 0 512 2048 2560 256 896 0 > $O/real_kernel_points.txt 2>&1
