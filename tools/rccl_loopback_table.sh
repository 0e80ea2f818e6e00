# usage: [LOOP_GRIDS="2x1x1 ..."] [LOOP_PADS="default 0"] bash tools/rccl_loopback_table.sh   (one box: single-device bench + loop-back runs)
mkdir -p gpurun_out/r2t; rm -f gpurun_out/r2t/loop_*.json; export TMPDIR=/tmp RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1
python bench.py --steps 200 --warmup 40 --cpu-seconds 0 2>/dev/null > gpurun_out/r2t/single.json
i=0
for g in ${LOOP_GRIDS:-2x1x1 2x2x1 2x2x2}; do for pad in ${LOOP_PADS:-default 0 default 0}; do
  i=$((i+1)); export MASTER_PORT=$((29720+i))
  if [ "$pad" = default ]; then unset LUDWIG_VIEW_PAD_BLOCKS; else export LUDWIG_VIEW_PAD_BLOCKS=$pad; fi
  timeout -k 10 300 python tests/_rccl_loopback_worker.py $g ${LOOP_NB:-32} 240 gpurun_out/r2t/loop_${g}_pad${pad}_$i.json nocompare >/dev/null 2>&1 || exit 1
done; done
unset LUDWIG_VIEW_PAD_BLOCKS
python bench.py --steps 200 --warmup 40 --cpu-seconds 0 2>/dev/null > gpurun_out/r2t/single2.json
python tools/loop_table.py
