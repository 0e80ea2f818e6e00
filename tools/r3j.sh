#!/bin/bash
O=gpurun_out/r3j
mkdir -p $O
export TMPDIR=/tmp RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29911
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_rccl_loopback.py -m gpu -q > $O/tests.log 2>&1; tail -3 $O/tests.log
for c in 1x1x2:32 1x2x4:64,32,16; do g=${c%%:*}; nb=${c##*:}; export MASTER_PORT=$((MASTER_PORT+1))
  LUDWIG_HALO_TRACE=1 timeout -k 10 200 python tests/_rccl_loopback_worker.py $g $nb 60 $O/trace_$g.json nocompare > /dev/null 2> $O/host_trace_$g.txt
  grep ludwig_halo_exchange $O/host_trace_$g.txt | tail -5
done
cd /tmp
for c in 1x2x4:64,32,16 2x2x2:32; do g=${c%%:*}; nb=${c##*:}; export MASTER_PORT=$((MASTER_PORT+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/$O/kt_$g -o t --output-format csv -- python3 $R/tests/_rccl_loopback_worker.py $g $nb 60 $R/$O/prof_$g.json nocompare > $R/$O/prof_$g.log 2>&1
  echo "exit $?"; grep -E "k_pack|k_unpack|rccl" $R/$O/kt_$g/t_kernel_stats.csv | cut -c1-160
  rm -f $R/$O/kt_$g/t_kernel_trace.csv
done
export MASTER_PORT=$((MASTER_PORT+1))
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/pmc_fetch -o p --output-format csv -- python3 $R/tests/_rccl_loopback_worker.py 1x2x4 64,32,16 24 $R/$O/pmc.json nocompare > $R/$O/pmc_fetch.log 2>&1
export MASTER_PORT=$((MASTER_PORT+1))
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/pmc_write -o p --output-format csv -- python3 $R/tests/_rccl_loopback_worker.py 1x2x4 64,32,16 24 $R/$O/pmc.json nocompare > $R/$O/pmc_write.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for which in ("fetch", "write"):
    v = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r3j/pmc_{which}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "octets" in r["Kernel_Name"]:
                v[(r["Kernel_Name"][:22], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, a in v.items():
        a = sorted(a); print(which, k, "calls", len(a), "values (KiB) min/median/max", a[0], a[len(a)//2], a[-1])
PY
rm -rf $O/pmc_fetch/*kernel_trace.csv $O/pmc_write/*kernel_trace.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt_wing -o t --output-format csv -- python3 $R/tools/case_speed.py wing 200 > $O/wing_prof.log 2>&1; head -14 $O/kt_wing/t_kernel_stats.csv | cut -c1-170; rm -f $O/kt_wing/t_kernel_trace.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt_sphere -o t --output-format csv -- python3 $R/tools/case_speed.py re266k 400 > $O/sphere_prof.log 2>&1; head -14 $O/kt_sphere/t_kernel_stats.csv | cut -c1-170; rm -f $O/kt_sphere/t_kernel_trace.csv
