# usage (GPU box): bash tools/post_readers_ab.sh   - RCCL loop-back of the 4-rank layout with a body on every cut (Bouzidi cells on both sides: f_post halo),
# per-rank step with f_post_collision stored in every block (round 2: LUDWIG_FULL_POST_COLLISION=1) against the rows with a reader (ludwig_level_add_post_collision_readers)
O=gpurun_out/prab; mkdir -p $O; export TMPDIR=/tmp RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1
i=0
for rep in 1 2; do for mode in every_block rows; do
  i=$((i+1)); export MASTER_PORT=$((29760+i))
  if [ $mode = every_block ]; then export LUDWIG_FULL_POST_COLLISION=1; else unset LUDWIG_FULL_POST_COLLISION; fi
  timeout -k 10 300 python tests/_rccl_loopback_worker.py 1x2x2 32 240 $O/${mode}_$rep.json nocompare spheres > /dev/null 2> $O/err_$i.txt || { tail -5 $O/err_$i.txt; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/${mode}_$rep.json')); print('$mode', {k: d[k] for k in d if 'ms' in k or k in ('bouzidi_cells','f_post_halo_elements','peers')})"
done; done
