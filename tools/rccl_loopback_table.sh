# usage: [LOOP_CASES="1x1x2:32 1x2x2:32 1x2x4:64,32,16"] [LOOP_TRANSPORTS="native torch"] bash tools/rccl_loopback_table.sh
# one box: single-device bench + RCCL loop-back runs of the weak-scaling layouts (grid:blocks per brick), per transport
O=gpurun_out/r3t
mkdir -p $O; rm -f $O/loop_*.json; export TMPDIR=/tmp RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1
python bench.py --steps 200 --warmup 40 --cpu-seconds 0 2>/dev/null > $O/single.json
i=0
for c in ${LOOP_CASES:-1x1x2:32 1x2x2:32 1x2x4:64,32,16}; do for tr in ${LOOP_TRANSPORTS:-native torch native}; do
  i=$((i+1)); export MASTER_PORT=$((29720+i)); g=${c%%:*}; nb=${c##*:}
  LOOPBACK_TRANSPORT=$tr timeout -k 10 300 python tests/_rccl_loopback_worker.py $g $nb 240 $O/loop_${g}_${tr}_$i.json nocompare >/dev/null 2>$O/loop_$i.err || { tail -5 $O/loop_$i.err; exit 1; }
done; done
python bench.py --steps 200 --warmup 40 --cpu-seconds 0 2>/dev/null > $O/single2.json
python tools/loop_table.py $O
