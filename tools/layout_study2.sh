#!/bin/bash
O=gpurun_out/r3b
mkdir -p $O
for i in 1 2 3; do
  tools/stridebench 32 20 0 32768 32768 1 | tail -1 | sed 's/^/pop-major 64.0 MiB: /'
  tools/stridebench 32 20 3 | tail -2
  tools/stridebench 32 20 5 | tail -2
  tools/stridebench 16 80 3 | tail -1 | sed 's/^/nb16 /'
  tools/stridebench 16 80 5 | tail -1 | sed 's/^/nb16 /'
  tools/stridebench 24 40 3 | tail -1 | sed 's/^/nb24 /'
  tools/stridebench 24 40 5 | tail -1 | sed 's/^/nb24 /'
done > $O/layouts_plane_major.txt 2>&1
cat $O/layouts_plane_major.txt
