#!/bin/bash
O=gpurun_out/r3o
mkdir -p $O
export TMPDIR=/tmp RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29700
python bench.py --steps 200 --warmup 40 --cpu-seconds 0 2>/dev/null > $O/single.json; python3 -c "import json; d=json.load(open('$O/single.json')); print('single device', d['ms_per_step'])"
for rep in 1 2; do
for c in 1x1x2:32 1x2x4:64,32,16; do g=${c%%:*}; nb=${c##*:}
for cu in 8 0 2; do
  export MASTER_PORT=$((MASTER_PORT+1))
  LUDWIG_COMM_RESERVED_CUS=$cu timeout -k 10 300 python tests/_rccl_loopback_worker.py $g $nb 240 $O/l.json nocompare 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$g reserved CUs $cu:', round(d['ms_per_step_wall'],4), 'ms per step, exchange span', round(d['exchange_ms_median_after_first'],3))"
done; done; done
