#!/bin/bash
# experiment: a stepping kernel capped at 4 waves per SIMD (104 VGPRs) instead of a CU mask, so that RCCL's kernels find room on every CU
O=gpurun_out/r3t
mkdir -p $O
export TMPDIR=/tmp RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29600
V=$PWD/tools/variants/lib_vgpr104.so
python bench.py --steps 200 --warmup 40 --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('single device, product library:', d['ms_per_step'])"
LUDWIG_HIP_LIB=$V python bench.py --steps 200 --warmup 40 --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('single device, 4-waves variant :', d['ms_per_step'])"
for rep in 1 2; do
for c in 1x1x2:32 1x2x4:64,32,16; do g=${c%%:*}; nb=${c##*:}
for cfg in "product:8" "product:0" "variant:0" "variant:8"; do
  lib=${cfg%%:*}; cu=${cfg##*:}; export MASTER_PORT=$((MASTER_PORT+1))
  if [ $lib = variant ]; then export LUDWIG_HIP_LIB=$V; else unset LUDWIG_HIP_LIB; fi
  LUDWIG_COMM_RESERVED_CUS=$cu timeout -k 10 300 python tests/_rccl_loopback_worker.py $g $nb 240 $O/l.json nocompare 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$g $lib library, reserved CUs $cu:', round(d['ms_per_step_wall'],4), 'ms per step, exchange span', round(d['exchange_ms_median_after_first'],3))"
done; done; done
