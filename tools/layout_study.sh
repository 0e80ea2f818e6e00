#!/bin/bash
# Round 3: population-major (stride swept) against block-major storage in the product's schedule, one box, alternating.
O=gpurun_out/r3b
mkdir -p $O
for i in 1 2 3; do
  tools/stridebench 32 20 0 32768 32768 1 | tail -1 | sed 's/^/pop-major 64.0 MiB: /'
  tools/stridebench 32 20 0 33280 33280 1 | tail -1 | sed 's/^/pop-major 65.0 MiB: /'
  tools/stridebench 32 20 0 34816 34816 1 | tail -1 | sed 's/^/pop-major 68.0 MiB: /'
  tools/stridebench 32 20 3 | tail -2
  tools/stridebench 32 20 4 | tail -2
done > $O/layouts_nb32.txt 2>&1
for i in 1 2; do
  tools/stridebench 16 80 0 4096 4096 1 | tail -1 | sed 's/^/nb16 pop-major 8 MiB: /'
  tools/stridebench 16 80 0 4648 4648 1 | tail -1 | sed 's/^/nb16 pop-major 9.08 MiB (bad): /'
  tools/stridebench 16 80 3 | tail -2
  tools/stridebench 24 40 0 13824 13824 1 | tail -1 | sed 's/^/nb24 pop-major 27 MiB: /'
  tools/stridebench 24 40 0 16512 16512 1 | tail -1 | sed 's/^/nb24 pop-major 32.25 MiB (bad): /'
  tools/stridebench 24 40 3 | tail -2
  tools/stridebench 64 4 0 262144 262144 1 | tail -1 | sed 's/^/nb64 pop-major 512 MiB: /'
  tools/stridebench 64 4 4 | tail -2
done > $O/layouts_other.txt 2>&1
cat $O/layouts_nb32.txt $O/layouts_other.txt
