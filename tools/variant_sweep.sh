#!/bin/bash
# time each library variant (separate processes, same device, interleaved rounds)
for r in 1 2; do
for v in "$@"; do
  LUDWIG_HIP_LIB=$PWD/tools/variants/$v.so python tools/order_sweep.py 256 pxcd_1x4_yxz,block_planes 2 2>/dev/null | grep -E "pxcd|block_planes" | sed "s/^/$v r$r /"
done
done
