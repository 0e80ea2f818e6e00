# usage (GPU box): bash tools/nested_loopback_in_stream_ab.sh - the 2-level RCCL loop-back (tests/_rccl_loopback_worker.py nested) with every level's
# exchange on the level's own stream (MultiLevelRunner's rule for small levels) against on the plan's stream under the interior part
O=gpurun_out/nlab; mkdir -p $O; export TMPDIR=/tmp RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 LOOPBACK_TRANSPORT=native
i=0
for g in ${NLAB_CASES:-2x1x1:4 4x1x1:4 2x1x1:12}; do for rep in 1 2; do for mode in in_stream overlap; do
  i=$((i+1)); export MASTER_PORT=$((29800+i))
  if [ $mode = overlap ]; then export LUDWIG_HALO_IN_STREAM_BELOW=0; else unset LUDWIG_HALO_IN_STREAM_BELOW; fi
  timeout -k 10 300 python tests/_rccl_loopback_worker.py ${g%%:*} ${g##*:} 200 $O/r_$i.json nested > /dev/null 2> $O/err_$i.txt || { tail -5 $O/err_$i.txt; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/r_$i.json')); print('$g $mode', 'blocks', d['blocks'], 'owned', d['owned'], 'in_stream', d['exchange_in_stream'], 'ms per coarse step %.4f' % d['ms_per_coarse_step_wall'], 'identical', all(d['identical'].values()))"
done; done; done
