#!/bin/bash
# loop-back table (native / torch transports) + the profiled command whose exit crashed in round 2 (gpurun_out/r2l/prof4.log)
O=gpurun_out/r3i
mkdir -p $O
export TMPDIR=/tmp RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29811
bash tools/rccl_loopback_table.sh > $O/loop_table.txt 2>&1; cat $O/loop_table.txt
cd /tmp
for tr in native torch; do
  export MASTER_PORT=$((MASTER_PORT+1))
  LOOPBACK_TRANSPORT=$tr timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace_$tr -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tests/_rccl_loopback_worker.py 2x2x2 32 30 $GRAFT_REPO_ROOT/$O/loop_prof_$tr.json nocompare > $GRAFT_REPO_ROOT/$O/prof_$tr.log 2>&1
  echo "exit code of the profiled $tr loop-back worker: $?" | tee -a $GRAFT_REPO_ROOT/$O/exit_codes.txt
done
cd $GRAFT_REPO_ROOT
for tr in native torch; do
  f=$(ls $O/trace_$tr/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/loopback_${tr}_kernel_stats.csv
  rm -rf $O/trace_$tr/*/*kernel_trace.csv
done
head -12 $O/loopback_native_kernel_stats.csv | cut -c1-180
