"""Turn rocprofv3 CSV output (kernel trace + PMC passes) into the small text/JSON summaries committed under profiles/."""
import csv, glob, json, os, re, sys, collections

def kernel_stats(trace_dir):
    rows = []
    for f in glob.glob(os.path.join(trace_dir, "**", "*_kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in agg.values())
    out = []
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        out.append({"kernel": k, "calls": len(v), "total_ns": sum(v), "avg_ns": sum(v) / len(v), "min_ns": min(v), "max_ns": max(v),
                    "pct": 100.0 * sum(v) / total})
    return out

def pmc(dirs, kernel_substr):
    vals = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                # the RHO_ONLY replay launch (fifth template argument true) is a different kernel: keep it out of the averages
                if kernel_substr in r["Kernel_Name"] and not re.search(r"true, (true|false)>\(lw::SCParams\)", r["Kernel_Name"]):
                    vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in vals.items()}

if __name__ == "__main__":
    trace_dir, pmc_root, out_prefix, cells = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    ks = kernel_stats(trace_dir)
    with open(out_prefix + "_kernel_stats.txt", "w") as f:
        f.write("rocprofv3 --kernel-trace --stats summary (per kernel, ns)\n")
        f.write(f"{'kernel':70s} {'calls':>6s} {'avg_ns':>12s} {'min_ns':>12s} {'max_ns':>12s} {'pct':>7s}\n")
        for k in ks:
            f.write(f"{k['kernel'][:70]:70s} {k['calls']:6d} {k['avg_ns']:12.0f} {k['min_ns']:12d} {k['max_ns']:12d} {k['pct']:7.2f}\n")
    c = pmc(glob.glob(os.path.join(pmc_root, "pmc_*")), "k_stream_collide")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from open_ludwig_amd import build as build_mod
    import datetime
    res = {"cells_per_launch": cells, "counters_per_launch": c, "source_digest": build_mod.source_digest(),
           "captured": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%d %H:%M UTC"),
           "kernel": "void lw::k_stream_collide_xrun<4, false, false, false, false, false>(lw::SCParams) at 256^3, library default order, rho store elided"}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B read requests
        # at 64 B -> double it; WRITE_SIZE is exact. Separate --pmc passes (TCC has 4 slots: FETCH 3 + WRITE 2 do not fit).
        rd = c["FETCH_SIZE"] * 1024 * 2
        wr = c["WRITE_SIZE"] * 1024
        res.update({"read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                    "read_bytes_per_cell": rd / cells, "write_bytes_per_cell": wr / cells,
                    "algorithmic_bytes_per_cell": 216, "note": "fabric-side (TCC_EA) bytes: Infinity-Cache hits are included, so this is an upper bound on HBM traffic"})
    json.dump(res, open(out_prefix + "_traffic.json", "w"), indent=1)
    print(open(out_prefix + "_kernel_stats.txt").read())
    print(json.dumps(res, indent=1))
