# usage (GPU box): bash tools/wing_pmc.sh  - fabric bytes per cell update of the shipped wing's kernels (rocprofv3 PMC, separate FETCH_SIZE / WRITE_SIZE passes)
O=gpurun_out/wpmc; mkdir -p $O; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c -d $O/$c -o p --output-format csv -- python3 tools/run_wing.py $O/run_$c 4 4 shipped > $O/$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
threads = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/wpmc/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            key = (r["Kernel_Name"].replace("void ", "").split("(")[0], int(r["Grid_Size"]))
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel (threads per launch): launches, fabric read B / thread (FETCH_SIZE x 2 KiB: the gfx950 correction), write B / thread, sum")
for key, d in sorted(agg.items(), key=lambda kv: -kv[0][1]):
    if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d or key[1] < 100000: continue
    rd = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]) * 1024 * 2 / key[1]
    wr = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"]) * 1024 / key[1]
    print(f"{key[0][:70]:70s} {key[1]:10d}: {len(d['FETCH_SIZE']):4d}  read {rd:7.1f}  write {wr:7.1f}  sum {rd + wr:7.1f}")
PY
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
