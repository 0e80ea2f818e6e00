#!/bin/bash
# L2 / fabric counters of the 256^3 stream-collide launch for library variants (one rocprofv3 pass per variant).
# usage: tools/pmc_variants.sh <out_dir> <variant> [variant...]     (variants live in tools/variants/<name>.so)
out=$1; shift
export TMPDIR=/tmp
mkdir -p $out
for v in "$@"; do
  LUDWIG_HIP_LIB=$PWD/tools/variants/$v.so timeout -k 10 150 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $out/$v/a -o pmc --output-format csv -- python3 tools/profile_step.py 256 6 > $out/$v.a.log 2>&1 || { echo "$v failed"; exit 1; }
  LUDWIG_HIP_LIB=$PWD/tools/variants/$v.so timeout -k 10 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/$v/b -o pmc --output-format csv -- python3 tools/profile_step.py 256 6 > $out/$v.b.log 2>&1 || { echo "$v failed"; exit 1; }
  python3 - "$out/$v" "$v" <<'PY'
import csv, glob, sys
from collections import defaultdict
d, v = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_stream_collide" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
n = 256 ** 3
m = {k: sum(x) / len(x) for k, x in acc.items()}
# FETCH_SIZE is in KiB-like units of 1024 B tallied at half width on gfx950 for 128-B requests (guide: x2); WRITE_SIZE exact
print(v, "L2 req/cell hit %.3f miss %.3f (hit rate %.3f)" % (m["TCC_HIT_sum"] / n, m["TCC_MISS_sum"] / n, m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])),
      "fabric read B/cell %.1f" % (m["FETCH_SIZE"] * 1024 * 2 / n))
PY
done
