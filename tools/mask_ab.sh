mkdir -p gpurun_out/r2t; export TMPDIR=/tmp RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1
i=0
for cfg in "1x1x2 32" "1x2x4 64,32,16"; do set -- $cfg; for r in 32 0 32 0; do
  i=$((i+1)); export MASTER_PORT=$((29960+i)) LUDWIG_COMM_RESERVED_CUS=$r
  timeout -k 10 300 python tests/_rccl_loopback_worker.py $1 $2 240 /tmp/m_$i.json nocompare >/dev/null 2>&1 || exit 1
  python -c "
import json; d=json.load(open('/tmp/m_$i.json')); print('$1 $2 reserved $r:', round(d['ms_per_step_wall'],4), round(d['exchange_ms_median_after_first'],3))"
done; done
